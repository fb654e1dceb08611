"""oracle/zstd_oracle.c (the RFC 8878 restatement used to cross-check the device decoder stage by stage) against
libzstd itself: pinned on the reference's .zst fixtures and on streams of every level libzstd writes."""
import ctypes as C
import gzip
import os
import subprocess

import pytest

from zstd_util import compress, decompress_stream, fastq_text, skippable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zso():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libzstd_oracle.so"])
    o = C.CDLL(os.path.join(ROOT, "oracle", "libzstd_oracle.so"))
    o.zso_decompress.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.c_void_p]
    o.zso_xxh64.restype = C.c_uint64
    o.zso_xxh64.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]

    def dec(comp, cap):
        out = C.create_string_buffer(cap + 64)
        p = C.c_uint64(0)
        rc = o.zso_decompress(comp, len(comp), out, cap, C.byref(p), None, 0, None)
        return rc, out.raw[:p.value]

    o.dec = dec
    return o


def test_reference_fixtures(zso, golden_dir):
    text = gzip.decompress(open(os.path.join(golden_dir, "test.fastq.gz"), "rb").read())
    for f in ("test.fastq.zst", "test.fastq.zstd"):
        assert zso.dec(open(os.path.join(golden_dir, f), "rb").read(), 1000) == (0, text)
    fa = open(os.path.join(golden_dir, "test.fasta"), "rb").read()
    for f in ("test.fasta.zst", "test.fasta.zstd"):
        assert zso.dec(open(os.path.join(golden_dir, f), "rb").read(), 1000) == (0, fa)


def test_xxh64_known_answers(zso):
    # published XXH64 test vectors (seed 0): the empty input, and "a"
    assert zso.zso_xxh64(b"", 0, 0) == 0xEF46DB3751D8E999
    assert zso.zso_xxh64(b"a", 1, 0) == 0xD24EC4F1A98C6E5B


@pytest.mark.parametrize("level", [1, 3, 5, 9, 15, 19, -3])
def test_levels_against_libzstd(zso, level):
    fq = fastq_text(8000, 3)
    for d in (b"", b"x", b"\0" * 70000, os.urandom(150000), fq, fq[:1000] * 90):
        for ck in (False, True):
            comp = compress(d, level, ck)
            assert decompress_stream(comp) == (True, d)
            assert zso.dec(comp, len(d)) == (0, d)


def test_frames_and_corruption(zso):
    d = fastq_text(9000, 5)
    comp = compress(d[:50000], 3, True) + skippable(b"meta") + compress(d[50000:], 12, True, window_log=11, content_size=False)
    assert zso.dec(comp, len(d)) == (0, d)
    bad = bytearray(compress(d, 3, True))
    bad[len(bad) // 2] ^= 0x10
    assert not decompress_stream(bytes(bad))[0]
    assert zso.dec(bytes(bad), len(d))[0] != 0
    assert zso.dec(comp[:-3], len(d))[0] != 0
