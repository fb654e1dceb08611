"""Bounded-memory decode: compressed inputs (BGZF, single-member gzip, zstd) and FASTA are scanned as a stream of device
batches / decoded segments whose size does not depend on the input — what the reference gets from a BufReader +
DataFusion's convert_stream (rust/src/arrow_reader.rs:60-91, 116-153).  EXG_DEVICE_MEM_CAP_MB makes the batches small
enough for a test: inputs whose decoded size is >= 8x the cap must return the rows of the uncapped run (and of the oracle),
while the device memory the reader holds (exg_reader_stats_of) stays under the cap."""
import gzip
import hashlib

import pytest

pytestmark = pytest.mark.gpu

CAP_MB = 16


def _oracle_digest(table, names):
    """the digest ShardReader.digest() computes, from the oracle's rows (flat VARCHAR columns)"""
    hs = []
    for c in names:
        h = hashlib.blake2b(digest_size=16)
        for v in table.columns[c].to_list():
            h.update(b"\xff\x00NULL" if v is None else v)
            h.update(b"\x00")
        hs.append(h.digest())
    return table.n_rows, hashlib.blake2b(b"".join(hs), digest_size=16).hexdigest()


def _bgzf(data, block=65280, level=1):
    """BGZF framing (what bgzip / htslib write) around zlib members; level 1: the tests deflate > 100 MB inputs"""
    import struct
    import zlib
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        d = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" +
                   struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1) + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")   # + BGZF EOF marker


def _open(path, fmt, **kw):
    from exon_duckdb_amd.reader import ShardReader
    return ShardReader(str(path), fmt, **kw)


def _capped_and_not(monkeypatch, path, fmt, compression=None):
    """(rows, digest) uncapped; (rows, digest, stats) under the cap"""
    monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
    r = _open(path, fmt, compression=compression)
    free = r.digest()
    r.close()
    monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(CAP_MB))
    r = _open(path, fmt, compression=compression)
    capped = r.digest()
    st = r.stats()
    r.close()
    r = _open(path, fmt, compression=compression)
    n = r.count()
    st2 = r.stats()
    r.close()
    assert n == capped[0]
    for s in (st, st2):
        assert s["device_mem_cap"] == CAP_MB << 20
        assert s["device_bytes_peak"] <= CAP_MB << 20, {k: v >> 10 if "bytes" in k else v for k, v in s.items()}
        assert s["device_batches"] >= 8
    return free, capped, st


@pytest.fixture(scope="module")
def fastq_big(oracle):
    data = bytes(oracle.synth_fastq(332 * 420000))          # 139 MB: > 8 x 16 MiB
    assert len(data) >= 8 * (CAP_MB << 20)
    exp = oracle.fastq_parse(data, want_string_t=False)
    return data, _oracle_digest(exp, ["name", "description", "sequence", "quality_scores"])


def test_bgzf_larger_than_the_cap(gpu, oracle, fastq_big, tmp_path, monkeypatch):
    data, want = fastq_big
    p = tmp_path / "big.fastq.gz"
    p.write_bytes(_bgzf(data))
    free, capped, st = _capped_and_not(monkeypatch, p, "fastq")
    assert free == want, "the uncapped run differs from the oracle"
    assert capped == want, "the capped run differs from the oracle"
    assert st["decoded_segments"] >= 8


def test_single_member_gzip_larger_than_the_cap(gpu, oracle, fastq_big, tmp_path, monkeypatch):
    data, want = fastq_big
    p = tmp_path / "big1.fastq.gz"
    p.write_bytes(gzip.compress(data, 1, mtime=0))
    free, capped, st = _capped_and_not(monkeypatch, p, "fastq")
    assert free == want, "the uncapped run differs from the oracle"
    assert capped == want, "the capped run differs from the oracle"
    assert st["decoded_segments"] >= 8


def test_fasta_larger_than_the_cap(gpu, oracle, tmp_path, monkeypatch):
    """plain and bgzipped FASTA: batches of whole records, the open record at a batch's end carried into the next"""
    data = bytes(oracle.synth_fasta(82000, seed=5))
    assert len(data) >= 8 * (CAP_MB << 20), len(data)
    want = _oracle_digest(oracle.fasta_parse(data), ["id", "description", "sequence"])
    p = tmp_path / "big.fasta"
    p.write_bytes(data)
    free, capped, st = _capped_and_not(monkeypatch, p, "fasta")
    assert free == want, "the uncapped run differs from the oracle"
    assert capped == want, "the capped run differs from the oracle"
    pz = tmp_path / "big.fasta.gz"
    pz.write_bytes(_bgzf(data))
    free, capped, st = _capped_and_not(monkeypatch, pz, "fasta")
    assert free == want, "the uncapped run differs from the oracle"
    assert capped == want, "the capped run differs from the oracle"


def test_fasta_record_longer_than_a_batch(gpu, oracle, tmp_path, monkeypatch):
    """a FASTA record can be as long as the input: a batch that holds no complete record is widened until it does"""
    import random
    rng = random.Random(3)
    recs = []
    for i, n_lines in enumerate([3, 40000, 2, 90000, 1, 5]):
        recs.append(b">r%d some description\n" % i + b"".join(bytes(rng.choice(b"ACGT") for _ in range(60)) + b"\n" for _ in range(n_lines)))
    data = b"".join(recs)
    want = oracle.fasta_parse(data)
    p = tmp_path / "long.fasta"
    p.write_bytes(data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(256 << 10))
    r = _open(p, "fasta")
    rows = r.rows()
    r.close()
    assert [x[0] for x in rows] == want.columns["id"].to_list()
    assert [x[2] for x in rows] == want.columns["sequence"].to_list()
    pz = tmp_path / "long.fasta.gz"
    pz.write_bytes(gzip.compress(data, 1, mtime=0))
    r = _open(pz, "fasta")
    rows2 = r.rows()
    r.close()
    assert rows2 == rows


def test_vcf_gz_larger_than_the_cap(gpu, oracle, tmp_path, monkeypatch):
    data = bytes(oracle.synth_vcf(2800000))
    monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
    p = tmp_path / "big.vcf"
    p.write_bytes(data)
    r = _open(p, "vcf")
    want = r.digest()
    r.close()
    assert want[0] == 2800000 and len(data) >= 8 * (CAP_MB << 20)
    pz = tmp_path / "big.vcf.gz"
    pz.write_bytes(_bgzf(data))
    free, capped, st = _capped_and_not(monkeypatch, pz, "vcf")
    assert free == want, "the uncapped run differs from the oracle"
    assert capped == want, "the capped run differs from the oracle"


def test_record_that_spans_segments(gpu, oracle, tmp_path, monkeypatch):
    """a FASTQ record far longer than a decoded segment (and than the room a segment leaves in front of itself): the tail
    moves into a block of its own and the batch is widened"""
    long_seq = b"ACGT" * 300000      # 1.2 MB
    data = b"".join(b"@r%d d\n%s\n+\n%s\n" % (i, (long_seq if i % 7 == 3 else b"ACGTACGT"), (b"I" * len(long_seq) if i % 7 == 3 else b"IIIIIIII"))
                    for i in range(40))
    want = oracle.fastq_parse(data, want_string_t=False)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(128 << 10))
    for name, blob in (("s.fastq.gz", _bgzf(data)), ("t.fastq.gz", gzip.compress(data, 1, mtime=0))):
        p = tmp_path / name
        p.write_bytes(blob)
        r = _open(p, "fastq")
        rows = r.rows()
        r.close()
        assert [x[0] for x in rows] == want.columns["name"].to_list()
        assert [x[2] for x in rows] == want.columns["sequence"].to_list()
        assert [x[3] for x in rows] == want.columns["quality_scores"].to_list()


def test_zstd_larger_than_the_cap(gpu, oracle, fastq_big, tmp_path, monkeypatch):
    """one frame (what the zstd CLI writes), with a Content_Checksum, decoded in rounds: the frame's window, its repeat
    offsets and the blocks whose tables it repeats travel from round to round; the checksum is folded on a host thread"""
    from zstd_util import compress
    data, want = fastq_big
    p = tmp_path / "big.fastq.zst"
    p.write_bytes(compress(data, 3, True, window_log=17))
    free, capped, st = _capped_and_not(monkeypatch, p, "fastq")
    assert free == want, "the uncapped run differs from the oracle"
    assert capped == want, "the capped run differs from the oracle"
    assert st["decoded_segments"] >= 8


@pytest.mark.parametrize("level,window_log", [(1, 0), (3, 18), (9, 20), (19, 21), (-3, 0)])
def test_zstd_rounds_against_libzstd(gpu, oracle, tmp_path, monkeypatch, level, window_log):
    """frames and parts of frames in rounds of many sizes: multi-frame streams (some frames smaller than a round, some
    spanning several), skippable frames in between, with and without checksums / content sizes — the rows of the plain file"""
    from zstd_util import compress, skippable
    data = bytes(oracle.synth_fastq(332 * 24000))                 # 8 MB
    want = _oracle_digest(oracle.fastq_parse(data, want_string_t=False), ["name", "description", "sequence", "quality_scores"])
    cuts = [0, 1000, 1000 + 332 * 300 + 7, 3_000_000, 3_000_001, 6_500_000, len(data)]
    parts = []
    for i in range(len(cuts) - 1):
        parts.append(compress(data[cuts[i]:cuts[i + 1]], level, i % 2 == 0, window_log=window_log, content_size=i % 3 != 1))
        if i == 2:
            parts.append(skippable(b"between frames"))
    blob = b"".join(parts)
    p = tmp_path / "m.fastq.zst"
    p.write_bytes(blob)
    for batch in (128 << 10, 400_000, 1 << 20, 3 << 20):
        monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(batch))
        r = _open(p, "fastq")
        got = r.digest()
        n_seg = r.stats()["decoded_segments"]
        r.close()
        assert got == want, (level, window_log, batch)
        assert n_seg >= (len(data) // max(batch, 128 << 10)) // 2
        r = _open(p, "fastq")
        assert r.count() == want[0]
        r.close()


def test_zstd_checksum_of_a_frame_that_spans_rounds(gpu, oracle, tmp_path, monkeypatch):
    """a wrong Content_Checksum of a frame decoded in several rounds is reported — behind the frame's rows, like a streaming
    decoder reports it (the device hashes only frames that lie inside one round)"""
    from exon_duckdb_amd import ExgError
    from zstd_util import compress
    data = bytes(oracle.synth_fastq(332 * 12000))
    blob = bytearray(compress(data, 3, True, window_log=18))
    blob[-1] ^= 0x40                                               # the checksum's last byte
    p = tmp_path / "bad.fastq.zst"
    p.write_bytes(bytes(blob))
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(512 << 10))
    r = _open(p, "fastq")
    with pytest.raises(ExgError, match="checksum"):
        r.count()
    r.close()
    r = _open(p, "fastq")
    with pytest.raises(ExgError, match="checksum"):
        r.rows()
    r.close()


def _shard_rows(path, fmt, n_shards, **kw):
    rows, peaks = [], []
    for k in range(n_shards):
        r = _open(path, fmt, shard_index=k, shard_count=n_shards, **kw)
        rows.extend(r.rows())
        peaks.append(r.stats()["device_bytes_peak"])
        r.close()
    return rows, peaks


def test_zstd_shard_behind_large_frames_holds_a_halo_not_the_frames(gpu, oracle, tmp_path, monkeypatch):
    """three frames of ~45 MB of FASTQ each, three shards: a shard's halo is the whole frame in front of its own.  Only the
    newest bytes of it stay resident while the decoder works its way to the shard's first byte (advisor, round 3: everything
    from the stream's first byte was, and was copied again for every segment more)."""
    from zstd_util import compress
    data = bytes(oracle.synth_fastq(332 * 400000))
    third = len(data) // 3 // 332 * 332
    parts = [data[:third + 100], data[third + 100:2 * third + 7], data[2 * third + 7:]]   # frames cut inside records
    p = tmp_path / "frames.fastq.zst"
    p.write_bytes(b"".join(compress(x, 1, True) for x in parts))
    exp = oracle.fastq_parse(data, want_string_t=False)
    want = list(zip(*[exp.columns[c].to_list() for c in ["name", "description", "sequence", "quality_scores"]]))
    # (twice the other tests' cap: the 1 MiB halo a shard keeps is an allocation more than an unsharded reader makes; a frame in
    # front of a shard decodes to 44 MB — resident, it would not fit)
    monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(2 * CAP_MB))
    rows, peaks = _shard_rows(p, "fastq", 3)
    assert rows == want
    assert max(peaks) <= (2 * CAP_MB) << 20, [x >> 20 for x in peaks]


def test_bgzf_shard_whose_phase_is_counted_holds_no_prefix(gpu, oracle, tmp_path, monkeypatch):
    """reads of 0.4 - 2 MB behind a 64 KiB halo: the 4-line phase of a late shard cannot be told from the bytes around its cut
    — the newlines in front of its members are counted by a decoder of their own, segment by segment, under the cap (advisor,
    round 3: the whole decoded prefix had to be resident)."""
    import numpy as np
    from exon_duckdb_amd.testing.shapes import fastq_records
    lengths = np.random.default_rng(4).integers(400_000, 2_000_000, 60)
    data = bytes(fastq_records(lengths, seed=31))
    assert len(data) > 8 * (CAP_MB << 20)
    p = tmp_path / "huge_reads.fastq.gz"
    p.write_bytes(_bgzf(data))
    exp = oracle.fastq_parse(data, want_string_t=False)
    want = list(zip(*[exp.columns[c].to_list() for c in ["name", "description", "sequence", "quality_scores"]]))
    monkeypatch.setenv("EXG_SHARD_HALO", str(64 << 10))
    monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(4 * CAP_MB))   # (a 2 MB read x 2 + its window must fit a batch)
    rows, peaks = _shard_rows(p, "fastq", 5)
    assert rows == want
    assert max(peaks) <= (4 * CAP_MB) << 20, [x >> 20 for x in peaks]
