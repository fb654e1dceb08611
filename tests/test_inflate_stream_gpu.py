"""One big DEFLATE stream decoded in chunks (exg_inflate_stream: block finder, 16-bit-symbol chunk decode, window
propagation, marker resolution) against zlib: FASTQ / VCF / text / random-ish data at several compression levels
and chunk sizes, streams with stored and fixed blocks, truncated and corrupted streams, and the same files through
the reader (`read_fastq('x.fastq.gz')` with a single gzip member)."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def stream_inflate(lib, raw_deflate: bytes, chunk_bytes, pad_front=0):
    import torch
    from exon_duckdb_amd import device
    lib.exg_inflate_stream.restype = C.c_int
    lib.exg_inflate_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
    data = b"\xAA" * pad_front + raw_deflate + b"\x55" * 8   # a header in front, a trailer behind (like gzip)
    d_comp = device.upload(data)
    out = C.c_void_p()
    produced, consumed = C.c_uint64(0), C.c_uint64(0)
    rc = lib.exg_inflate_stream(C.c_void_p(d_comp.data_ptr()), pad_front, len(data) - pad_front, chunk_bytes, C.byref(out),
                                C.byref(produced), C.byref(consumed), device.stream_ptr())
    if rc != 0:
        return rc, None, 0
    n = produced.value
    host = (C.c_uint8 * max(n, 1))()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(host, out, n, 2) == 0
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipFree(out)
    return 0, bytes(host)[:n], consumed.value


def deflate(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return co.compress(payload) + co.flush()


def payloads(oracle):
    rng = np.random.default_rng(12)
    fq = bytes(oracle.synth_fastq(332 * 60000))                         # 20 MB of FASTQ-150
    vcf = bytes(oracle.synth_vcf(150000))
    text = (b"the quick brown fox jumps over the lazy dog; " * 7 + b"\n") * 40000
    words = np.array([b"alpha", b"beta", b"gamma", b"delta", b"ACGT", b"\t", b"\n", b"0.125", b"PASS", b"rs"], dtype=object)
    mix = b"".join(words[rng.integers(0, len(words), 2_000_000)])
    noisy = bytes(rng.integers(0, 256, 3_000_000, dtype=np.uint8)) + fq[:3_000_000]   # stored blocks, then dynamic
    # compressible but NOT text (dynamic blocks whose literals are control bytes): the block finder's text probe
    # finds no start, the search falls back to whole-block validation
    binary = (np.arange(3_000_000, dtype=np.uint32) // 7 * 2654435761 % 1000).astype(np.uint16).tobytes()
    # text with a few control / non-ASCII bytes sprinkled in (legal UTF-8 in a description, a stray 0x01)
    dirty = bytearray(fq[:6_000_000])
    for k in range(0, len(dirty), 50_021):
        dirty[k] = (0x01, 0xC3, 0xA9, 0x7F)[(k // 50_021) % 4]
    return {"fastq": fq, "vcf": vcf, "text": text, "mix": mix, "noisy": noisy, "binary": binary, "dirty": bytes(dirty)}


@pytest.fixture(scope="module")
def data(oracle):
    return payloads(oracle)


@pytest.mark.parametrize("name", ["fastq", "vcf", "text", "mix", "noisy", "binary", "dirty"])
@pytest.mark.parametrize("level", [1, 6, 9])
def test_stream_against_zlib(gpu, data, name, level):
    payload = data[name]
    comp = deflate(payload, level)
    for chunk, pad in ((1 << 20, 10), (200_000, 13)):
        rc, got, consumed = stream_inflate(gpu, comp, chunk, pad_front=pad)
        assert rc == 0, gpu.exg_last_error_message()
        assert got == payload, (name, level, chunk, len(got), len(payload))
        assert consumed == len(comp)


@pytest.mark.parametrize("seed", [1, 2])
def test_stream_with_flush_points(gpu, data, seed):
    """pigz-style streams: sync / full flushes (an empty stored block, byte alignment) between blocks of a few bytes to a few
    hundred KB — block ends in every lane of a speculative step, chunk ends next to empty stored blocks"""
    rng = np.random.default_rng(seed)
    payload = data["fastq"][:6_000_000] + data["text"][:1_000_000]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    out, pos = [], 0
    while pos < len(payload):
        n = int(rng.integers(1, 400)) if rng.integers(0, 3) == 0 else int(rng.integers(20_000, 300_000))
        out.append(co.compress(payload[pos:pos + n]))
        out.append(co.flush(zlib.Z_FULL_FLUSH if rng.integers(0, 4) == 0 else zlib.Z_SYNC_FLUSH))
        pos += n
    out.append(co.flush())
    comp = b"".join(out)
    for chunk, pad in ((1 << 20, 3), (150_000, 16)):
        rc, got, consumed = stream_inflate(gpu, comp, chunk, pad_front=pad)
        assert rc == 0, gpu.exg_last_error_message()
        assert got == payload
        assert consumed == len(comp)


def test_fixed_huffman_and_tiny_streams(gpu):
    for payload in (b"", b"a", b"hello hello hello hello", bytes(range(256)) * 300):
        comp = deflate(payload, 6, zlib.Z_FIXED)
        rc, got, consumed = stream_inflate(gpu, comp, 65536)
        assert rc == 0 and got == payload and consumed == len(comp)


def test_corrupt_and_truncated_streams_fail(gpu, data):
    comp = bytearray(deflate(data["fastq"][:4_000_000], 6))
    rc, _, _ = stream_inflate(gpu, bytes(comp[: len(comp) // 2]), 1 << 18)        # truncated: no final block
    assert rc != 0
    comp[len(comp) // 3] ^= 0x5A                                                    # a flipped byte in the middle
    rc, got, _ = stream_inflate(gpu, bytes(comp), 1 << 18)
    assert rc != 0 or got != data["fastq"][:4_000_000]                              # never silently "fine"


def test_reader_single_member_gzip(gpu, oracle, tmp_path, monkeypatch):
    """read_fastq on a plain `gzip` file (one member): count, rows, and the same answer as the per-member path."""
    from exon_duckdb_amd import table_function
    raw = bytes(oracle.synth_fastq_ragged(40000))
    p = tmp_path / "single.fastq.gz"
    p.write_bytes(gzip.compress(raw, 6, mtime=0))
    assert os.path.getsize(p) > (1 << 20)
    monkeypatch.setenv("EXG_STREAM_MIN_BYTES", str(1 << 16))
    monkeypatch.setenv("EXG_STREAM_CHUNK_BYTES", str(1 << 17))
    con = table_function.connect()
    rel = con.table_function("read_fastq", str(p))
    exp = oracle.fastq_parse(raw, want_string_t=False)
    want = list(zip(*[exp.columns[k].to_list() for k in ("name", "description", "sequence", "quality_scores")]))
    assert rel.count() == len(want) == 40000
    assert rel.fetchall() == want
    monkeypatch.setenv("EXG_NO_STREAM_INFLATE", "1")
    assert con.table_function("read_fastq", str(p)).fetchall(limit=2000) == want[:2000]   # one wavefront: slow but equal


def test_concatenated_big_members(gpu, oracle, tmp_path, monkeypatch):
    from exon_duckdb_amd import table_function
    raw = bytes(oracle.synth_fastq(332 * 30000))
    p = tmp_path / "two.fastq.gz"
    p.write_bytes(gzip.compress(raw[: 332 * 20000], 6, mtime=0) + gzip.compress(raw[332 * 20000:], 6, mtime=0))
    monkeypatch.setenv("EXG_STREAM_MIN_BYTES", str(1 << 16))
    con = table_function.connect()
    assert con.table_function("read_fastq", str(p)).count() == 30000


def test_cat_of_big_members_and_bgzf_blocks(gpu, oracle, tmp_path):
    """`cat a.gz b.gz`-style inputs: every big member of unknown size takes the chunked decode (a later one used to
    fall to ONE wavefront: 13 MB/s), sized BGZF blocks in between take the per-member kernel; the rows are the
    concatenation."""
    import struct

    from exon_duckdb_amd import table_function
    raw = bytes(oracle.synth_fastq_ragged(9000, seed=77))
    # cut at record boundaries so that every part is a FASTQ file of its own
    lines = raw.split(b"\n")
    rec_bytes = [sum(len(x) + 1 for x in lines[i:i + 4]) for i in range(0, len(lines) - (len(lines) % 4), 4)]
    cuts = [2500, 4000, 6500]
    offs = [0] + [sum(rec_bytes[:c]) for c in cuts] + [len(raw)]
    parts = [raw[offs[i]:offs[i + 1]] for i in range(4)]

    def bgzf(data):
        out = []
        for i in range(0, len(data), 65280):
            chunk = data[i:i + 65280]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            d = co.compress(chunk) + co.flush()
            out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" +
                       struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1) + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        return b"".join(out)

    blob = gzip.compress(parts[0], 6, mtime=0) + bgzf(parts[1]) + gzip.compress(parts[2], 6, mtime=0) + gzip.compress(parts[3], 6, mtime=0)
    assert all(len(gzip.compress(p, 6, mtime=0)) > (128 << 10) for p in (parts[0], parts[2], parts[3]))
    p = tmp_path / "cat.fastq.gz"
    p.write_bytes(blob)
    q = tmp_path / "cat.fastq"
    q.write_bytes(raw)
    con = table_function.connect()
    want = con.table_function("read_fastq", str(q)).fetchall()
    rel = con.table_function("read_fastq", str(p))
    assert rel.count() == len(want) == 9000
    assert rel.fetchall() == want


@pytest.mark.parametrize("seed", range(int(os.environ.get("EXG_GZ_FUZZ", "12"))))
def test_random_gzip_compositions(gpu, oracle, tmp_path, seed):
    """Files glued together from plain gzip members and runs of BGZF blocks of random sizes (around the 128 KiB
    switch between one wavefront and the chunked decode): rows and COUNT(*) equal the plain file's."""
    import struct

    from exon_duckdb_amd import table_function
    rng = np.random.default_rng(7000 + seed)
    n_parts = int(rng.integers(1, 7))
    recs = [int(rng.integers(1, 4000)) for _ in range(n_parts)]
    raw = bytes(oracle.synth_fastq(332 * sum(recs)))           # fixed-size records: any multiple of 332 is a cut
    blob, off = [], 0
    for k, n in enumerate(recs):
        part = raw[off:off + 332 * n]
        off += 332 * n
        if rng.integers(0, 2):
            block = int(rng.integers(500, 65280))
            for i in range(0, len(part), block):
                chunk = part[i:i + block]
                co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
                d = co.compress(chunk) + co.flush()
                blob.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" +
                            struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1) + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        else:
            blob.append(gzip.compress(part, int(rng.integers(1, 10)), mtime=0))
    p = tmp_path / "mix.fastq.gz"
    p.write_bytes(b"".join(blob))
    q = tmp_path / "mix.fastq"
    q.write_bytes(raw)
    con = table_function.connect()
    want = con.table_function("read_fastq", str(q)).fetchall()
    rel = con.table_function("read_fastq", str(p))
    assert rel.count() == len(want) == sum(recs), (seed, recs)
    assert rel.fetchall() == want, (seed, recs)
