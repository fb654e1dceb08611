"""Random FASTQ texts through the READER in random wrappings — plain, one gzip member (any zlib strategy), BGZF with random member
sizes, zstd at random levels / windows / frame cuts / checksums — in random round sizes, with and without a memory cap: the rows
of the oracle's parse of the plain text, whatever the wrapping and however the stream was cut into rounds (what a streaming
decoder in front of the reference's line readers gives: rust/src/arrow_reader.rs:60-91).  EXG_STREAM_SOAK scales the seeds
(default 16; a run of 400 at the end of a round)."""
import os
import random
import zlib

import pytest

from test_streaming_gpu import _bgzf, _open, _oracle_digest

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("EXG_STREAM_SOAK", "16"))
COLS = ["name", "description", "sequence", "quality_scores"]


def fastq_text(r):
    """records of many shapes: short reads, a few very long ones, descriptions or none, names that repeat their neighbours'"""
    out = bytearray()
    n = r.randint(1, 4000)
    base = r.randrange(10 ** 6)
    long_every = r.choice([0, 0, 50, 700])
    for i in range(n):
        ln = r.randint(1, 300)
        if long_every and i % long_every == long_every - 1:
            ln = r.randint(2000, 60000)
        seq = bytes(r.choice(b"ACGTN") for _ in range(ln)) if ln < 400 else (bytes(r.choice(b"ACGT") for _ in range(97)) * (ln // 97 + 1))[:ln]
        qual = bytes(33 + (j * 7 + i) % 41 for j in range(ln))
        name = b"@r%d.%d" % (base, i)
        if r.random() < 0.5:
            name += b" len=%d %s" % (ln, b"x" * r.randint(0, 20))
        out += name + b"\n" + seq + b"\n+\n" + qual + b"\n"
    return bytes(out)


def fasta_text(r):
    """records wrapped at a width of their own, a few of them far longer than a round, descriptions or none"""
    out = bytearray()
    for i in range(r.randint(1, 1500)):
        ln = r.choice([0, 1, 59, 60, 61]) if r.random() < 0.1 else r.randint(1, 2000)
        if r.random() < 0.01:
            ln = r.randint(100_000, 600_000)
        width = r.choice([60, 70, 80, 1, 7, 120, 10 ** 9])
        seq = (bytes(r.choice(b"ACGTN") for _ in range(min(ln, 211))) * (ln // 211 + 1))[:ln]
        out += b">s%d" % i + (b" d%d %s" % (i, b"y" * r.randint(0, 30)) if r.random() < 0.6 else b"") + b"\n"
        for o in range(0, ln, width):
            out += seq[o:o + width] + b"\n"
    return bytes(out)


def wrap(r, data, tmp_path, seed, ext="fastq"):
    from zstd_util import compress, skippable
    kind = r.choice(["plain", "gz", "bgzf", "zst", "zst", "zst"])
    if kind == "plain":
        p = tmp_path / f"s{seed}.{ext}"
        p.write_bytes(data)
    elif kind == "gz":
        co = zlib.compressobj(r.randint(1, 9), zlib.DEFLATED, 31, 8, r.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        p = tmp_path / f"s{seed}.{ext}.gz"
        p.write_bytes(co.compress(data) + co.flush())
    elif kind == "bgzf":
        p = tmp_path / f"s{seed}.{ext}.gz"
        p.write_bytes(_bgzf(data, block=r.choice([997, 8191, 32768, 65280]), level=r.randint(1, 9)))
    else:
        level = r.choice([1, 2, 3, 3, 5, 9, 15, 19, -1, -5])
        wl = r.choice([0, 0, 10, 12, 14, 17, 20])
        cuts = sorted({0, len(data)} | {r.randrange(len(data) + 1) for _ in range(r.choice([0, 0, 1, 3, 9]))})
        parts = []
        for a, b in zip(cuts, cuts[1:]):
            parts.append(compress(data[a:b], level, r.random() < 0.5, window_log=wl, content_size=r.random() < 0.7))
            if r.random() < 0.15:
                parts.append(skippable(b"x" * r.randint(0, 40)))
        p = tmp_path / f"s{seed}.{ext}.zst"
        p.write_bytes(b"".join(parts))
    return kind, p


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("seed0", range(0, N_SEEDS, 8))
def test_random_streams(gpu, oracle, tmp_path, monkeypatch, seed0):
    for seed in range(seed0, min(seed0 + 8, N_SEEDS)):
        r = random.Random(7000 + seed)
        data = fastq_text(r)
        want = _oracle_digest(oracle.fastq_parse(data, want_string_t=False), COLS)
        kind, p = wrap(r, data, tmp_path, seed)
        for mode in range(2):
            if r.random() < 0.35:
                monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(r.choice([16, 24, 64])))
                monkeypatch.delenv("EXG_DEVICE_BATCH_BYTES", raising=False)
            else:
                monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
                monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(r.choice([128 << 10, 300_000, 1 << 20, 4 << 20, 0])))
            rd = _open(p, "fastq")
            got = rd.digest()
            rd.close()
            assert got == want, (seed, kind, mode, len(data), dict((k, os.environ.get(k)) for k in ("EXG_DEVICE_MEM_CAP_MB", "EXG_DEVICE_BATCH_BYTES")))
        rd = _open(p, "fastq")
        assert rd.count() == want[0], (seed, kind)
        rd.close()
        os.unlink(p)


def _env(r, monkeypatch):
    if r.random() < 0.35:
        monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(r.choice([16, 24, 64])))
        monkeypatch.delenv("EXG_DEVICE_BATCH_BYTES", raising=False)
    else:
        monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
        monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(r.choice([128 << 10, 300_000, 1 << 20, 4 << 20, 0])))


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("seed0", range(0, N_SEEDS, 8))
def test_random_fasta_streams(gpu, oracle, tmp_path, monkeypatch, seed0):
    """FASTA: a record is as long as it is (batches are widened until they hold one), sequences are joined across their lines"""
    for seed in range(seed0, min(seed0 + 8, N_SEEDS)):
        r = random.Random(9000 + seed)
        data = fasta_text(r)
        want = _oracle_digest(oracle.fasta_parse(data), ["id", "description", "sequence"])
        kind, p = wrap(r, data, tmp_path, seed, "fasta")
        for mode in range(2):
            _env(r, monkeypatch)
            rd = _open(p, "fasta")
            got = rd.digest()
            rd.close()
            assert got == want, (seed, kind, mode, len(data), dict((k, os.environ.get(k)) for k in ("EXG_DEVICE_MEM_CAP_MB", "EXG_DEVICE_BATCH_BYTES")))
        os.unlink(p)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("seed0", range(0, N_SEEDS, 8))
def test_random_vcf_streams(gpu, tmp_path, monkeypatch, seed0):
    """VCF (flat, typed and nested columns): a wrapped file gives the digest of the plain file, however it is cut into rounds
    (the plain file against the oracle: tests/test_vcf_*.py, test_record_shapes_gpu.py)"""
    from exon_duckdb_amd.testing import shapes
    for seed in range(seed0, min(seed0 + 8, N_SEEDS)):
        r = random.Random(11000 + seed)
        data = shapes.vcf_lines(r.randint(1, 3000), r.choice([0, 0, 1, 3, 40, 120]), seed=seed, crlf_every=r.choice([0, 0, 7]))
        plain = tmp_path / f"p{seed}.vcf"
        plain.write_bytes(data)
        monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
        monkeypatch.delenv("EXG_DEVICE_BATCH_BYTES", raising=False)
        rd = _open(plain, "vcf")
        want = rd.rows()          # (every column as Python values: the nested ones too, whatever the chunking)
        rd.close()
        kind, p = wrap(r, data, tmp_path, seed, "vcf")
        for mode in range(2):
            _env(r, monkeypatch)
            rd = _open(p, "vcf")
            got = rd.rows()
            rd.close()
            assert len(got) == len(want), (seed, kind, mode)
            bad = next((i for i in range(len(want)) if repr(got[i]) != repr(want[i])), None)   # (repr: NaN == NaN)
            assert bad is None, (seed, kind, mode, bad, got[bad], want[bad], dict((k, os.environ.get(k)) for k in ("EXG_DEVICE_MEM_CAP_MB", "EXG_DEVICE_BATCH_BYTES")))
        os.unlink(p)
        if plain.exists():
            os.unlink(plain)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("seed0", range(0, N_SEEDS, 8))
def test_random_streams_at_the_arrow_boundary(gpu, oracle, tmp_path, monkeypatch, seed0):
    # the reference's own boundary (new_reader -> Arrow C stream): a text file's batch is handed on while its buffers still travel
    # (round 6: ABatch::landed, two arenas in turn) — small device batches, so that a file is dozens of them, in every wrapping,
    # a slow consumer every now and then (the producer thread then runs a whole batch ahead); every value must be the oracle's
    import time
    from exon_duckdb_amd.arrow import new_reader
    from test_arrow_stream_gpu import fasta_rows, fastq_rows
    for seed in range(seed0, min(seed0 + 8, N_SEEDS)):
        r = random.Random(9000 + seed)
        fasta = r.random() < 0.4
        data = fasta_text(r) if fasta else fastq_text(r)
        want = fasta_rows(oracle, data) if fasta else fastq_rows(oracle, data)
        kind, p = wrap(r, data, tmp_path, seed, ext="fasta" if fasta else "fastq")
        monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
        monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(r.choice([64 << 10, 128 << 10, 300_000, 1 << 20, 0])))
        slow = r.random() < 0.3
        rows = []
        for b in new_reader(str(p), "fasta" if fasta else "fastq", batch_size=r.choice([2048, 128, 64])):
            rows.extend(b.to_pylist())
            if slow and r.random() < 0.02:
                time.sleep(0.002)
        assert rows == want, (seed, kind, fasta, len(data), os.environ.get("EXG_DEVICE_BATCH_BYTES"))
        os.unlink(p)
