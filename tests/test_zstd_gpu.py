"""zstd on the device (exg_zstd_decode) against libzstd: every compression level, block type and table mode the encoder
produces, multi-frame / skippable frames / checksums / missing content size / small windows, the reference's own .zst
fixtures, and corrupted or truncated streams (an error wherever libzstd gives one)."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from zstd_util import compress, decompress_stream, fastq_text, skippable

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def zstd_decode(lib, comp: bytes):
    """-> (rc, bytes | message)"""
    from exon_duckdb_amd import device
    d_comp = device.upload(comp)
    host = C.create_string_buffer(comp, len(comp))
    out = C.c_void_p()
    produced = C.c_uint64(0)
    rc = lib.exg_zstd_decode(C.cast(host, C.c_void_p), C.c_void_p(d_comp.data_ptr()), len(comp), C.byref(out), C.byref(produced),
                             device.stream_ptr())
    if rc != 0:
        return rc, lib.exg_last_error_message().decode()
    n = produced.value
    buf = (C.c_uint8 * max(n, 1))()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(buf, out, n, 2) == 0
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipFree(out)
    return 0, bytes(buf)[:n]


def payloads():
    r = random.Random(5)
    mixed = bytearray()
    for i in range(40):
        mixed += os.urandom(r.randint(1, 5000)) if i % 3 == 0 else fastq_text(r.randint(1, 400), i) if i % 3 == 1 else bytes(
            [r.randint(0, 255)]) * r.randint(1, 9000)
    return {
        "empty": b"",
        "one": b"a",
        "short": b"abc" * 5,
        "zeros": b"\0" * 100000,
        "ramp": bytes(range(256)) * 100,
        "random": os.urandom(300000),
        "fastq_small": fastq_text(3000),
        "fastq": fastq_text(20000, 7),
        "runs": b"A" * 70000 + os.urandom(1000) + b"A" * 70000,
        "acgtn": bytes(r.choice(b"ACGTN") for _ in range(400000)),
        "vcf_lines": b"chr1\t12345\trs99\tA\tG\t50.0\tPASS\tDP=10;AF=0.5\n" * 8000,
        "mixed": bytes(mixed),
    }


@pytest.fixture(scope="module")
def data():
    return payloads()


@pytest.mark.parametrize("name", list(payloads().keys()))
def test_levels_and_checksums(gpu, data, name):
    d = data[name]
    for level in (1, 2, 3, 4, 5, 7, 9, 12, 15, 19, -1, -5):
        for ck in (False, True):
            comp = compress(d, level, ck)
            rc, out = zstd_decode(gpu, comp)
            assert rc == 0, (name, level, ck, out)
            assert out == d, (name, level, ck, len(out), len(d))


def test_reference_fixtures(gpu):
    for f, first in (("test.fastq.zst", b"@SEQ_ID"), ("test.fastq.zstd", b"@SEQ_ID"), ("test.fasta.zst", b">a desc"), ("test.fasta.zstd", b">a desc")):
        comp = open(os.path.join(GOLDEN, f), "rb").read()
        ok, want = decompress_stream(comp)
        assert ok
        rc, out = zstd_decode(gpu, comp)
        assert rc == 0, out
        assert out == want and out.startswith(first)
    # the fixture holds the same text as test.fastq.gz (record 1 without a description), not test.fastq's
    import gzip
    plain = gzip.decompress(open(os.path.join(GOLDEN, "test.fastq.gz"), "rb").read())
    rc, out = zstd_decode(gpu, open(os.path.join(GOLDEN, "test.fastq.zst"), "rb").read())
    assert out == plain


def test_multi_frame_skippable_window_no_content_size(gpu, data):
    d = fastq_text(30000, 3)
    comp = (compress(d[:100000], 3, True) + skippable(b"hello", 3) + compress(d[100000:], 19, True, window_log=12, content_size=False)
            + compress(b"", 3) + skippable(b"") + compress(b"tail", 1, True))
    ok, want = decompress_stream(comp)
    assert ok and want == d + b"tail"
    rc, out = zstd_decode(gpu, comp)
    assert rc == 0, out
    assert out == want
    comp = compress(d, 5, True, window_log=10, content_size=False)
    rc, out = zstd_decode(gpu, comp)
    assert rc == 0 and out == d
    # many small frames (what pzstd / a chunked writer produces)
    parts = [d[i:i + 7001] for i in range(0, len(d), 7001)]
    comp = b"".join(compress(p, 1 + i % 7, i % 2 == 0) for i, p in enumerate(parts))
    rc, out = zstd_decode(gpu, comp)
    assert rc == 0 and out == d
    # only skippable frames / nothing at all
    assert zstd_decode(gpu, skippable(b"x" * 100)) == (0, b"")
    assert zstd_decode(gpu, b"") == (0, b"")


def test_big_frame_many_blocks(gpu):
    d = fastq_text(150000, 11)  # ~45 MB: hundreds of blocks, repeated tables, treeless literals
    for level, wl in ((1, 0), (3, 0), (6, 17), (19, 0)):
        comp = compress(d, level, True, window_log=wl)
        rc, out = zstd_decode(gpu, comp)
        assert rc == 0, out
        assert out == d, (level, wl)


def test_corrupt_streams_fail_like_libzstd(gpu, data):
    d = data["fastq"]
    r = random.Random(9)
    comp = compress(d, 3, True)
    # truncations
    for cut in (1, 3, 4, 5, 9, 20, len(comp) // 2, len(comp) - 5, len(comp) - 1):
        ok, _ = decompress_stream(comp[:cut])
        assert not ok
        rc, msg = zstd_decode(gpu, comp[:cut])
        assert rc != 0, cut
    # bit flips: the device must fail wherever libzstd fails, and agree byte for byte where libzstd accepts
    n_err = 0
    for trial in range(200):
        b = bytearray(comp)
        k = r.randrange(len(b))
        b[k] ^= 1 << r.randrange(8)
        ok, want = decompress_stream(bytes(b))
        rc, out = zstd_decode(gpu, bytes(b))
        if ok:
            assert rc == 0 and out == want, (trial, k)
        else:
            n_err += 1
            assert rc != 0, (trial, k, want)
    assert n_err > 150  # the checksum catches what the entropy stages let through
    # garbage, wrong magic, a dictionary id, a window beyond libzstd's default limit
    assert zstd_decode(gpu, b"not zstd at all")[0] != 0
    assert zstd_decode(gpu, b"\x28\xb5\x2f\xfd\x01\x58\x07\x00\x00\x00")[0] != 0  # Dictionary_ID flag set
    huge = compress(b"x" * 1000, 3, False, content_size=False)
    b = bytearray(huge)
    assert not (b[4] & 0x20)  # not single-segment: byte 5 is the window descriptor
    b[5] = (18 << 3)          # window log 28
    assert not decompress_stream(bytes(b))[0]
    assert zstd_decode(gpu, bytes(b))[0] != 0


# ---- through the reader (exg_open) and the reference's FFI (new_reader) -----------------------------------------------

def test_reader_fastq_zst_equals_plain(gpu, tmp_path, monkeypatch):
    from exon_duckdb_amd.reader import ShardReader
    text = fastq_text(40000, 21)
    plain = tmp_path / "r.fastq"
    plain.write_bytes(text)
    # one frame; many frames + a skippable one; a frame without content size and a 4 KiB window
    variants = {
        "one.fastq.zst": compress(text, 3, True),
        "many.fastq.zst": b"".join(compress(text[i:i + 300001], 1 + (i // 300001) % 5, True) for i in range(0, len(text), 300001)) + skippable(b"idx"),
        "win.fastq.zst": compress(text, 9, False, window_log=12, content_size=False),
    }
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(1 << 20))  # several device batches over the decoded bytes
    want = ShardReader(str(plain), "fastq").rows()
    assert len(want) == 40000
    for name, comp in variants.items():
        p = tmp_path / name
        p.write_bytes(comp)
        assert ShardReader(str(p), "fastq").count() == 40000, name
        assert ShardReader(str(p), "fastq").rows() == want, name
    # explicit compression on a name that says nothing
    q = tmp_path / "noext"
    q.write_bytes(variants["one.fastq.zst"])
    assert ShardReader(str(q), "fastq", compression="zstd").count() == 40000
    # shards of a zstd input go by frames (tests/test_reader_shards_gpu.py): a file of one frame is one shard's
    counts = [ShardReader(str(tmp_path / "one.fastq.zst"), "fastq", shard_index=i, shard_count=2).count() for i in range(2)]
    assert sorted(counts) == [0, 40000]


def test_reader_vcf_and_fasta_zst(gpu, golden_dir, tmp_path):
    from exon_duckdb_amd.reader import ShardReader
    vcf = open(os.path.join(golden_dir, "vcf", "index.vcf"), "rb").read()
    p = tmp_path / "index.vcf.zst"
    p.write_bytes(compress(vcf, 5, True))
    assert ShardReader(str(p), "vcf").count() == 621
    assert ShardReader(str(p), "vcf").rows() == ShardReader(os.path.join(golden_dir, "vcf", "index.vcf"), "vcf").rows()
    fa = b"".join(b">s%d some text\n" % i + b"ACGTTGCA" * (5 + i % 40) + b"\nGGCC\n" for i in range(5000))
    q = tmp_path / "x.fasta.zst"
    q.write_bytes(compress(fa, 3, True))
    (tmp_path / "x.fasta").write_bytes(fa)
    assert ShardReader(str(q), "fasta").rows() == ShardReader(str(tmp_path / "x.fasta"), "fasta").rows()


def test_new_reader_zst(gpu, golden_dir, tmp_path):
    from exon_duckdb_amd import arrow
    t = arrow.new_reader(os.path.join(golden_dir, "test.fastq.zst"), "fastq").read_all()
    assert t.num_rows == 2 and t.column("name").to_pylist() == ["SEQ_ID", "SEQ_ID2"]
    t = arrow.new_reader(os.path.join(golden_dir, "test.fasta.zstd"), "fasta", compression="zstd").read_all()
    assert t.num_rows == 2 and t.column("id").to_pylist() == ["a", "b"]
    vcf = open(os.path.join(golden_dir, "vcf", "index.vcf"), "rb").read()
    p = tmp_path / "index.vcf.zst"
    p.write_bytes(compress(vcf, 3, True))
    a = arrow.new_reader(str(p), "vcf").read_all()
    b = arrow.new_reader(os.path.join(golden_dir, "vcf", "index.vcf"), "vcf").read_all()
    assert a.equals(b) and a.num_rows == 621


def test_reader_corrupt_zst_is_an_error(gpu, tmp_path):
    from exon_duckdb_amd import ExgError
    from exon_duckdb_amd.reader import ShardReader
    comp = bytearray(compress(fastq_text(5000, 2), 3, True))
    comp[len(comp) // 3] ^= 0x40
    p = tmp_path / "bad.fastq.zst"
    p.write_bytes(bytes(comp))
    assert not decompress_stream(bytes(comp))[0]
    with pytest.raises(ExgError):
        ShardReader(str(p), "fastq").count()
    (tmp_path / "cut.fastq.zst").write_bytes(bytes(compress(fastq_text(5000, 2), 3, True)[:-7]))
    with pytest.raises(ExgError):
        ShardReader(str(tmp_path / "cut.fastq.zst"), "fastq").count()


def test_big_frame_checksum_is_verified_on_the_host(gpu, tmp_path):
    """A frame above the device's XXH64 limit (64 MiB of content; the hash is a serial recurrence): its decoded bytes travel
    back in pieces and a host thread hashes them.  A right checksum passes; one wrong bit in it is libzstd's error — from
    exg_zstd_decode at once, from a reader once the file's last batch has been handed out (the rows in front of it are
    delivered, as a streaming decoder would)."""
    from exon_duckdb_amd import ExgError, device
    from exon_duckdb_amd.reader import ShardReader
    n_rec = 230_000
    d = bytes(device.synth_fastq(332 * n_rec)[: 332 * n_rec].cpu().numpy())
    assert len(d) > (64 << 20)
    comp = compress(d, 1, True)
    rc, out = zstd_decode(gpu, comp)
    assert rc == 0 and out == d
    bad = bytearray(comp)
    bad[-1] ^= 0x01  # the frame's last four bytes are the checksum
    assert not decompress_stream(bytes(bad))[0]
    rc, msg = zstd_decode(gpu, bytes(bad))
    assert rc != 0 and "checksum" in msg, msg
    good_p, bad_p = tmp_path / "good.fastq.zst", tmp_path / "bad.fastq.zst"
    good_p.write_bytes(comp)
    bad_p.write_bytes(bytes(bad))
    assert ShardReader(str(good_p), "fastq").count() == n_rec
    with pytest.raises(ExgError, match="checksum"):
        ShardReader(str(bad_p), "fastq").count()
    rd = ShardReader(str(bad_p), "fastq")
    from exon_duckdb_amd.table_function import Chunk
    seen, rc = 0, 0
    while True:
        ch = Chunk()
        rc = rd._l.exg_next_chunk(rd._r, C.byref(ch))
        if rc != 0 or ch.n_rows == 0:
            break
        seen += int(ch.n_rows)
        rd._l.exg_release_chunk(rd._r, C.byref(ch))
    assert rc != 0 and seen == n_rec
    with pytest.raises(ExgError, match="checksum"):
        rd._fail(rc)


def test_soak_fixtures_and_seeds(gpu):
    """frames whose sequence tables are in RLE mode (a mode libzstd picks when every sequence of a block has the same code: small
    windows make such blocks) — the two fixtures are seeds 482 and 253 of tools/zstd_soak.py, which the decoder refused until
    round 4 ("Data corruption detected (sequences)": the RLE byte was not skipped) — and a run of the soak's seeds"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    for f in sorted(os.listdir(os.path.join(GOLDEN, "zstd_soak"))):
        comp = open(os.path.join(GOLDEN, "zstd_soak", f), "rb").read()
        ok, want = decompress_stream(comp)
        assert ok
        rc, out = zstd_decode(gpu, comp)
        assert rc == 0, (f, out)
        assert out == want, f
    import zstd_soak
    total, failed = zstd_soak.run(list(range(40)) + [33, 236, 253, 300, 482])
    assert not failed, failed
