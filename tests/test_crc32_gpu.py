"""CRC-32 of inflated bytes on the device (exg_crc32_segments / exg_crc32_members, zlib.crc32 as the checker) and what
the reader does with it: every gzip member's CRC-32 and ISIZE are verified against its trailer, like flate2's GzDecoder /
noodles-bgzf behind rust/src/arrow_reader.rs:60-91 — a payload that still decodes but no longer matches is an error."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_segments_against_zlib(gpu):
    import torch
    from exon_duckdb_amd import device
    rng = np.random.default_rng(4)
    data = bytes(rng.integers(0, 256, 1_500_000, dtype=np.uint8))
    d = device.upload(data)
    segs = [(0, 0), (0, 1), (1, 2), (5, 63), (7, 64), (100, 65), (3, 1000), (11, 4095), (64, 65279), (0, 65280), (13, 65535), (1, 65536),
            (2, 65537), (17, 200_000), (0, len(data))]
    segs += [(int(rng.integers(0, 1_000_000)), int(rng.integers(0, 300_000))) for _ in range(200)]
    arr = np.array(segs, dtype=np.uint64).reshape(-1, 2)
    d_segs = torch.from_numpy(arr.view(np.int64)).cuda()
    d_crc = torch.zeros(len(segs), dtype=torch.int32, device="cuda")
    gpu.exg_crc32_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    assert gpu.exg_crc32_segments(C.c_void_p(d.data_ptr()), C.c_void_p(d_segs.data_ptr()), len(segs), C.c_void_p(d_crc.data_ptr()),
                                  device.stream_ptr()) == 0
    got = d_crc.cpu().numpy().view(np.uint32)
    for (off, ln), c in zip(segs, got):
        assert int(c) == zlib.crc32(data[off:off + ln]), (off, ln)
    gpu.exg_crc32_combine.restype = C.c_uint32
    gpu.exg_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
    a, b = data[:70001], data[70001:170000]
    assert gpu.exg_crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)


def bgzf(data, block=65280):
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1)
                   + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def count(path, **kw):
    from exon_duckdb_amd.reader import ShardReader
    return ShardReader(path, "fastq", **kw).count()


def test_reader_verifies_every_trailer(gpu, oracle, tmp_path):
    from exon_duckdb_amd import ExgError
    text = bytes(oracle.synth_fastq(332 * 3000))
    # (1) BGZF: a wrong CRC in one member's trailer, then a wrong ISIZE
    good = bgzf(text, 5000)
    (tmp_path / "ok.fastq.gz").write_bytes(good)
    assert count(str(tmp_path / "ok.fastq.gz")) == 3000
    assert count(str(tmp_path / "ok.fastq.gz"), shard_index=1, shard_count=3) > 0
    first_len = struct.unpack("<H", good[16:18])[0] + 1
    bad = bytearray(good)
    bad[first_len - 8] ^= 1                                   # CRC32 of member 0
    (tmp_path / "crc.fastq.gz").write_bytes(bytes(bad))
    with pytest.raises(ExgError, match="checksum"):
        count(str(tmp_path / "crc.fastq.gz"))
    with pytest.raises(ExgError, match="checksum"):          # the shard that owns member 0
        count(str(tmp_path / "crc.fastq.gz"), shard_index=0, shard_count=3)
    # (2) a payload byte changed to another literal of the same code length still decodes — only the checksum tells
    single = bytearray(gzip.compress(text, 6, mtime=0))
    (tmp_path / "one.fastq.gz").write_bytes(bytes(single))
    assert count(str(tmp_path / "one.fastq.gz")) == 3000
    tr = bytearray(single)
    tr[-8] ^= 0x40                                            # the big member's CRC32 (decoded in chunks: exg_inflate_stream)
    (tmp_path / "one_crc.fastq.gz").write_bytes(bytes(tr))
    with pytest.raises(ExgError, match="checksum"):
        count(str(tmp_path / "one_crc.fastq.gz"))
    tr = bytearray(single)
    tr[-2] ^= 0x01                                            # its ISIZE
    (tmp_path / "one_isize.fastq.gz").write_bytes(bytes(tr))
    with pytest.raises(ExgError):
        count(str(tmp_path / "one_isize.fastq.gz"))
    # (3) small members of unknown size (the one-wavefront path): `cat a.gz b.gz`, second trailer wrong
    a, b = gzip.compress(text[:332 * 40], 6, mtime=0), gzip.compress(text[332 * 40:332 * 90], 6, mtime=0)
    (tmp_path / "cat.fastq.gz").write_bytes(a + b)
    assert count(str(tmp_path / "cat.fastq.gz")) == 90
    bb = bytearray(a + b)
    bb[-6] ^= 0x10
    (tmp_path / "cat_crc.fastq.gz").write_bytes(bytes(bb))
    with pytest.raises(ExgError, match="checksum"):
        count(str(tmp_path / "cat_crc.fastq.gz"))
    # flip bits inside the deflate payload of a BGZF file: an error whenever the bytes differ (decode error or checksum)
    rng = np.random.default_rng(8)
    n_err = 0
    for trial in range(30):
        x = bytearray(good)
        k = int(rng.integers(18, len(good) - 40))
        x[k] ^= 1 << int(rng.integers(0, 8))
        (tmp_path / "flip.fastq.gz").write_bytes(bytes(x))
        try:
            ref = gzip.decompress(bytes(x))
        except Exception:
            ref = None
        if ref is None:
            n_err += 1
            with pytest.raises(ExgError):
                count(str(tmp_path / "flip.fastq.gz"))
        else:
            assert ref == text and count(str(tmp_path / "flip.fastq.gz")) == 3000   # (a flip in a header field zlib ignores)
    assert n_err >= 20


def test_highly_compressible_small_members_are_not_corrupt(gpu, tmp_path):
    """`cat a.gz b.gz` of small members with a ratio far above 8 (ADVICE): the one-wavefront path bounds its output by
    DEFLATE's 1032:1, not by a guessed ratio."""
    rec = b"@r\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"
    a, b = gzip.compress(rec * 20000, 9, mtime=0), gzip.compress(rec * 500, 9, mtime=0)
    assert len(a) < 100_000 and len(rec) * 20000 / len(a) > 100
    (tmp_path / "r.fastq.gz").write_bytes(a + b)
    assert count(str(tmp_path / "r.fastq.gz")) == 20500
