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


def wrap(r, data, tmp_path, seed):
    from zstd_util import compress, skippable
    kind = r.choice(["plain", "gz", "bgzf", "zst", "zst", "zst"])
    if kind == "plain":
        p = tmp_path / f"s{seed}.fastq"
        p.write_bytes(data)
    elif kind == "gz":
        co = zlib.compressobj(r.randint(1, 9), zlib.DEFLATED, 31, 8, r.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        p = tmp_path / f"s{seed}.fastq.gz"
        p.write_bytes(co.compress(data) + co.flush())
    elif kind == "bgzf":
        p = tmp_path / f"s{seed}.fastq.gz"
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
        p = tmp_path / f"s{seed}.fastq.zst"
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
