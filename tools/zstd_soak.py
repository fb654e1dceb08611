#!/usr/bin/env python3
"""Random payloads through the device zstd decoder against the bytes that were compressed (libzstd wrote the frames):
short-period repeats (matches that overlap themselves and each other inside 64 output elements), text, noise, long runs —
what k_zst_exec's element-parallel passes have to get right (and, with small windows, blocks whose sequence tables are in
RLE mode: seeds 33, 236, 253, 300, 482 found the dropped byte of exg_zstd.hip's seq_table_from).  ZSOAK_SEEDS (300),
ZSOAK_FROM (0), ZSOAK_DUMP (path prefix: a failing seed's frame is written there); run on the GPU box."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import torch
from exon_duckdb_amd import device, load_library
from zstd_util import compress

lib = load_library()
lib.exg_zstd_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]


def decode(comp):
    d_comp = device.upload(comp)
    host = C.create_string_buffer(comp, len(comp))
    out, produced = C.c_void_p(), C.c_uint64(0)
    rc = lib.exg_zstd_decode(C.cast(host, C.c_void_p), C.c_void_p(d_comp.data_ptr()), len(comp), C.byref(out), C.byref(produced), device.stream_ptr())
    if rc:
        return rc, lib.exg_last_error_message().decode()
    n = produced.value
    buf = (C.c_uint8 * max(n, 1))()
    assert hip.hipMemcpy(buf, out, n, 2) == 0
    hip.hipFree(out)
    return 0, bytes(buf)[:n]


def payload(r):
    out = bytearray()
    for _ in range(r.randint(1, 60)):
        kind = r.randrange(7)
        if kind == 0:    # a pattern of period 1 .. 70 repeated: matches that overlap themselves
            pat = bytes(r.randrange(256) for _ in range(r.randint(1, 70)))
            out += pat * r.randint(2, 4000 // len(pat) + 2)
        elif kind == 1:  # noise
            out += r.randbytes(r.randint(1, 20000))
        elif kind == 2:  # a long run
            out += bytes([r.randrange(256)]) * r.randint(1, 200000)
        elif kind == 3:  # DNA: short matches found far back
            out += bytes(r.choice(b"ACGT") for _ in range(r.randint(100, 60000)))
        elif kind == 4:  # records whose names repeat their neighbours'
            k = r.randrange(10 ** 6)
            for i in range(r.randint(1, 300)):
                out += b"@read.%d.%d len=%d\n" % (k, i, r.randint(50, 150)) + bytes(r.choice(b"ACGT") for _ in range(r.randint(20, 160))) + b"\n+\n" + bytes(r.randint(33, 73) for _ in range(40)) + b"\n"
        elif kind == 5 and out:  # a copy of something earlier, any distance
            a = r.randrange(len(out))
            out += out[a:a + r.randint(1, 5000)]
        else:            # short repeats interleaved: sources inside the same 64 elements as their copies
            for _ in range(r.randint(1, 400)):
                out += bytes(r.randrange(4) + 65 for _ in range(r.randint(1, 6))) * r.randint(1, 5)
    return bytes(out)


def case(seed):
    """-> (payload, compressed frame) of a seed: libzstd wrote the frame (level, window, checksum, content size from the seed)"""
    r = random.Random(seed)
    d = payload(r)
    level = r.choice([1, 1, 2, 3, 3, 3, 5, 7, 9, 12, 19, -1, -5])
    wl = r.choice([0, 0, 0, 10, 12, 17, 20])
    return d, compress(d, level, r.random() < 0.5, window_log=wl, content_size=r.random() < 0.7), (level, wl)


def run(seeds, dump=None):
    """decodes every seed's frame on the device; -> (bytes checked, [(seed, level, window_log, what)] of the failures)"""
    total, failed = 0, []
    for seed in seeds:
        d, comp, (level, wl) = case(seed)
        rc, out = decode(comp)
        if rc or out != d:
            if dump:
                open(dump + f".{seed}.zst", "wb").write(comp)
            failed.append((seed, level, wl, out if rc else "bytes differ at %s" % next((i for i in range(min(len(out), len(d))) if out[i] != d[i]), min(len(out), len(d)))))
        total += len(d)
    return total, failed


if __name__ == "__main__":
    n = int(os.environ.get("ZSOAK_SEEDS", "300"))
    first = int(os.environ.get("ZSOAK_FROM", "0"))
    total, failed = run(range(first, first + n), os.environ.get("ZSOAK_DUMP"))
    for f in failed:
        print("FAILED", *f, flush=True)
    print(f"zstd soak: seeds {first}..{first + n - 1}: {len(failed)} failed, {total / 1e6:.1f} MB")
    sys.exit(1 if failed else 0)
