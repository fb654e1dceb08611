"""libzstd (the system's libzstd.so.1, through ctypes) as the checker of the device zstd decoder — tests only.
The reference decodes .zst through zstd 0.12.3 = libzstd 1.5.2 (rust/Cargo.lock:3875-3876); the frame format is fixed
(RFC 8878), so any libzstd produces streams the reference accepts and gives the verdicts it would give."""
import ctypes as C
import random

_z = None


def lib():
    global _z
    if _z is None:
        z = C.CDLL("libzstd.so.1")
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        z.ZSTD_isError.restype = C.c_uint
        z.ZSTD_isError.argtypes = [C.c_size_t]
        z.ZSTD_getErrorName.restype = C.c_char_p
        z.ZSTD_getErrorName.argtypes = [C.c_size_t]
        z.ZSTD_createCCtx.restype = C.c_void_p
        z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        z.ZSTD_CCtx_setParameter.restype = C.c_size_t
        z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        z.ZSTD_compress2.restype = C.c_size_t
        z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.ZSTD_createDStream.restype = C.c_void_p
        z.ZSTD_freeDStream.argtypes = [C.c_void_p]
        z.ZSTD_initDStream.argtypes = [C.c_void_p]
        z.ZSTD_initDStream.restype = C.c_size_t
        z.ZSTD_decompressStream.restype = C.c_size_t
        z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _z = z
    return _z


# ZSTD_cParameter values (zstd.h, stable API)
C_LEVEL, C_WINDOWLOG, C_CONTENTSIZE, C_CHECKSUM = 100, 101, 200, 201


def compress(data: bytes, level=3, checksum=False, window_log=0, content_size=True) -> bytes:
    z = lib()
    cc = z.ZSTD_createCCtx()
    try:
        for k, v in ((C_LEVEL, level), (C_CHECKSUM, int(checksum)), (C_CONTENTSIZE, int(content_size))):
            assert not z.ZSTD_isError(z.ZSTD_CCtx_setParameter(cc, k, v))
        if window_log:
            assert not z.ZSTD_isError(z.ZSTD_CCtx_setParameter(cc, C_WINDOWLOG, window_log))
        cap = z.ZSTD_compressBound(len(data))
        out = C.create_string_buffer(cap)
        n = z.ZSTD_compress2(cc, out, cap, data, len(data))
        assert not z.ZSTD_isError(n), z.ZSTD_getErrorName(n)
        return out.raw[:n]
    finally:
        z.ZSTD_freeCCtx(cc)


class _Buf(C.Structure):
    _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


def decompress_stream(comp: bytes):
    """(ok, bytes) the way a streaming consumer (async-compression's ZstdDecoder over ZSTD_decompressStream) sees the
    input: concatenated and skippable frames are read through; any libzstd error, or an input that ends inside a frame,
    is a failure."""
    z = lib()
    ds = z.ZSTD_createDStream()
    try:
        z.ZSTD_initDStream(ds)
        src = C.create_string_buffer(comp, len(comp))
        inb = _Buf(C.cast(src, C.c_void_p), len(comp), 0)
        chunk = C.create_string_buffer(1 << 17)
        out = bytearray()
        ret = 0
        while True:
            outb = _Buf(C.cast(chunk, C.c_void_p), len(chunk), 0)
            ret = z.ZSTD_decompressStream(ds, C.byref(outb), C.byref(inb))
            if z.ZSTD_isError(ret):
                return False, z.ZSTD_getErrorName(ret).decode()
            out += chunk.raw[:outb.pos]
            if inb.pos == inb.size and outb.pos < outb.size:
                break
        if ret != 0:
            return False, "truncated"
        return True, bytes(out)
    finally:
        z.ZSTD_freeDStream(ds)


def skippable(payload: bytes, nibble=0) -> bytes:
    return (0x184D2A50 + nibble).to_bytes(4, "little") + len(payload).to_bytes(4, "little") + payload


def fastq_text(n_records, seed=1, lo=50, hi=150) -> bytes:
    r = random.Random(seed)
    out = []
    for k in range(n_records):
        L = r.randint(lo, hi)
        out.append("@SYN%012d %d:N:0:ACGT\n%s\n+\n%s\n" % (k, k % 4, "".join(r.choice("ACGT") for _ in range(L)),
                                                            "".join(chr(r.randint(33, 73)) for _ in range(L))))
    return "".join(out).encode()
