"""Device inflate (exg_inflate_members) against zlib: every DEFLATE block type, long codes, overlapping
matches, window-sized distances, BGZF-style multi-member streams, the reference's .gz fixtures."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from exon_duckdb_amd import abi

pytestmark = pytest.mark.gpu


def index_members(lib, data: bytes):
    """Host framing (product code, exg_gzip_index) with the open-ended continuation loop left to the caller."""
    arr = np.frombuffer(data, np.uint8)
    cap = max(16, len(data) // 18 + 4)
    members = (abi.InflateMember * cap)()
    n = C.c_uint64(0)
    total = C.c_uint64(0)
    open_ended = C.c_int(0)
    rc = lib.exg_gzip_index(arr.ctypes.data, len(data), 0, members, cap, C.byref(n), C.byref(total), C.byref(open_ended))
    assert rc == 0, lib.exg_last_error_message()
    return [members[i] for i in range(n.value)], total.value, bool(open_ended.value)


def inflate_gpu(lib, data: bytes, members, total_out):
    import torch
    from exon_duckdb_amd import device

    d_comp = device.upload(data)
    d_out = torch.zeros(total_out + 64, dtype=torch.uint8, device="cuda")
    marr = (abi.InflateMember * len(members))(*members)
    d_members = torch.frombuffer(bytearray(bytes(marr)), dtype=torch.uint8).cuda()
    d_status = torch.zeros(len(members) * 24, dtype=torch.uint8, device="cuda")
    device.check(lib.exg_inflate_members(C.c_void_p(d_comp.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                         C.c_void_p(d_members.data_ptr()), C.c_void_p(d_status.data_ptr()),
                                         len(members), device.stream_ptr()))
    torch.cuda.synchronize()
    st = np.frombuffer(d_status.cpu().numpy().tobytes(), dtype=np.dtype([("code", "<u4"), ("pad", "<u4"), ("produced", "<u8"), ("consumed", "<u8")]))
    return d_out.cpu().numpy(), st


def roundtrip(gpu, payload: bytes, gz: bytes):
    members, total, open_ended = index_members(gpu, gz)
    out, st = inflate_gpu(gpu, gz, members, total)
    assert (st["code"] == 0).all(), st
    got = b"".join(out[m.out_off:m.out_off + int(s["produced"])].tobytes() for m, s in zip(members, st))
    assert got == payload
    return members, st


def bgzf(payload: bytes, block=65280, level=6):
    out = []
    for i in range(0, max(len(payload), 1), block):
        chunk = payload[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(raw) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                   + raw + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # BGZF EOF marker
    return b"".join(out)


def fastq_like(n, seed=1):
    rng = np.random.default_rng(seed)
    recs = []
    for k in range(n):
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), 150).tobytes()
        q = rng.integers(33, 74, 150, dtype=np.uint8).tobytes()
        recs.append(b"@SYN%012d %d:N:0:ACGT\n" % (k, k % 4) + seq + b"\n+\n" + q + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_single_member_levels(gpu, level):
    payload = fastq_like(300)
    gz = gzip.compress(payload, compresslevel=level, mtime=0)      # level 0 = stored blocks
    members, st = roundtrip(gpu, payload, gz)
    assert len(members) == 1
    assert int(st["consumed"][0]) + 8 == members[0].comp_size      # deflate stream ends right before CRC32 + ISIZE


def test_fixed_huffman_and_tiny_inputs(gpu):
    for payload in [b"", b"a", b"hello hello hello hello", bytes(range(256)) * 3]:
        co = zlib.compressobj(9, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
        gz = co.compress(payload) + co.flush()
        roundtrip(gpu, payload, gz)
        roundtrip(gpu, payload, gzip.compress(payload, mtime=0))


def test_overlapping_matches_and_far_distances(gpu):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    payload = b"x" * 5000 + b"ab" * 4000 + a + a[:300] + b"tail" + a + b"\0" * 70000
    roundtrip(gpu, payload, gzip.compress(payload, mtime=0))


def short_match_payload(seed, n):
    """literal runs of 0-4 random bytes between copies of 3-18 bytes from chosen distances: next to the copy itself (overlap),
    inside the step that emits it, around the edge of the decoder's 2 KiB ring, around its 1 KiB flush lag, far back, and
    at the window's limit — what zlib turns into runs of short matches (many per step of the decoder)"""
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGTN\n@+FI#:0123456789", dtype=np.uint8)
    out = bytearray(alphabet[rng.integers(0, len(alphabet), 64)].tobytes())
    bands = [(1, 20), (20, 200), (900, 1100), (1980, 2120), (3000, 9000), (32700, 32768)]
    while len(out) < n:
        out += alphabet[rng.integers(0, len(alphabet), int(rng.integers(0, 5)))].tobytes()
        lo, hi = bands[int(rng.integers(0, len(bands)))]
        d = int(rng.integers(lo, hi + 1))
        if d > len(out):
            continue
        ln = int(rng.integers(3, 19))
        for _ in range(ln):          # (byte by byte: a copy may overlap itself)
            out.append(out[-d])
    return bytes(out[:n])


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_runs_of_short_matches_at_every_distance(gpu, seed):
    """the batched copy of a step's independent short matches (exg_inflate_core.hpp, `fast`) against zlib: sources in the
    ring, in HBM, across the ring's edge; copies fed by a match of the same step; more matches in a step than slots"""
    from test_inflate_stream_gpu import deflate, stream_inflate
    payload = short_match_payload(seed, 700_000)
    for level in (1, 6, 9):
        roundtrip(gpu, payload, bgzf(payload, block=[65280, 30000, 4096][seed % 3], level=level))
        roundtrip(gpu, payload[:65000], gzip.compress(payload[:65000], level, mtime=0))
    # the chunked decoder (16-bit symbols: a byte, or a byte of the window in front of the chunk) takes the same step
    raw = deflate(payload, 6)
    rc, got, consumed = stream_inflate(gpu, raw, 40_000 + 1000 * seed)
    assert rc == 0 and got == payload and consumed == len(raw)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_many_tiny_blocks(gpu, seed):
    """a block end every few bytes (sync / full flushes at random places, all three block types): the token that ends a
    speculative step — end-of-block — falls into every window and every lane of the step, with block headers, empty stored
    blocks and byte alignment right behind it"""
    rng = np.random.default_rng(seed)
    text = fastq_like(120) + b"\0" * 3000 + bytes(rng.integers(0, 256, 4000, dtype=np.uint8)) + b"ACGT" * 2000
    for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY):
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, strategy)
        out, pos = [], 0
        while pos < len(text):
            n = int(rng.integers(1, 200))
            out.append(co.compress(text[pos:pos + n]))
            out.append(co.flush(zlib.Z_FULL_FLUSH if rng.integers(0, 4) == 0 else zlib.Z_SYNC_FLUSH))
            pos += n
        out.append(co.flush())
        roundtrip(gpu, text, b"".join(out))


def test_long_codes(gpu):
    # a skewed byte distribution forces code lengths > 10 bits (secondary canonical decode)
    rng = np.random.default_rng(9)
    p = np.array([2.0 ** -min(i, 40) for i in range(256)])
    payload = rng.choice(256, 200000, p=p / p.sum()).astype(np.uint8).tobytes()
    co = zlib.compressobj(9, zlib.DEFLATED, 31, 9, zlib.Z_HUFFMAN_ONLY)
    roundtrip(gpu, payload, co.compress(payload) + co.flush())


def test_gzip_header_fields(gpu):
    payload = b"ACGT" * 1000
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(payload) + raw.flush()
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16]) + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 4) + b"XY\0\0" + b"name.fq\0" + b"a comment\0"
    roundtrip(gpu, payload, hdr + body + struct.pack("<II", zlib.crc32(payload), len(payload)))


@pytest.mark.parametrize("n_records", [10, 2000, 20000])
def test_bgzf_members_in_parallel(gpu, n_records):
    payload = fastq_like(n_records, seed=n_records)
    gz = bgzf(payload)
    members, st = roundtrip(gpu, payload, gz)
    assert len(members) == (len(payload) + 65279) // 65280 + 1
    assert gzip.decompress(gz) == payload


def test_reference_gz_fixtures(gpu, golden_dir):
    for name in ["test.fastq.gz", "test.fasta.gz", "fasta/copy-a.fasta.gz", "vcf/index.vcf.gz"]:
        with open(os.path.join(golden_dir, name), "rb") as f:
            gz = f.read()
        roundtrip(gpu, gzip.decompress(gz), gz)


def test_corrupt_stream_is_reported(gpu):
    payload = fastq_like(100)
    gz = bytearray(gzip.compress(payload, mtime=0))
    gz[40] ^= 0x55
    members, total, _ = index_members(gpu, bytes(gz))
    out, st = inflate_gpu(gpu, bytes(gz), members, total)
    got = out[:int(st["produced"][0])].tobytes()
    assert st["code"][0] != 0 or got != payload      # never silently the right answer by accident
