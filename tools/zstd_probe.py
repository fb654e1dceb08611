"""zstd decode on the device at size: FASTQ-150 (device generator) compressed by libzstd on the host, decoded by
exg_zstd_decode; stage times come from EXG_TRACE=1.  usage: zstd_probe.py [MB ...] [--level L] [--frames K]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch  # noqa: E402

from exon_duckdb_amd import device, load_library  # noqa: E402
from zstd_util import compress  # noqa: E402

args = sys.argv[1:]
level = int(args[args.index("--level") + 1]) if "--level" in args else 3
frames = int(args[args.index("--frames") + 1]) if "--frames" in args else 1
sizes = [int(a) for a in args if a.isdigit() and args[args.index(a) - 1] not in ("--level", "--frames")] or [64, 512]
lib = load_library()
hip = C.CDLL("libamdhip64.so")
hip.hipFree.argtypes = [C.c_void_p]
for mb in sizes:
    n = mb * 1000 * 1000 // 332 * 332
    d = device.synth_fastq(n)
    torch.cuda.synchronize()
    host = bytes(d[:n].cpu().numpy())
    t0 = time.time()
    per = (n // frames + 331) // 332 * 332
    comp = b"".join(compress(host[i:i + per], level, True) for i in range(0, n, per))
    t_c = time.time() - t0
    d_comp = device.upload(comp)
    hbuf = C.create_string_buffer(comp, len(comp))
    for rep in range(3):
        out = C.c_void_p()
        produced = C.c_uint64(0)
        torch.cuda.synchronize()
        t0 = time.time()
        rc = lib.exg_zstd_decode(C.cast(hbuf, C.c_void_p), C.c_void_p(d_comp.data_ptr()), len(comp), C.byref(out), C.byref(produced), device.stream_ptr())
        dt = time.time() - t0
        assert rc == 0, lib.exg_last_error_message()
        assert produced.value == n
        if rep == 0:
            got = torch.empty(n, dtype=torch.uint8, device="cuda")
            hipc = hip.hipMemcpy
            hipc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hipc(C.c_void_p(got.data_ptr()), out, n, 3) == 0
            assert torch.equal(got, d[:n]), "decoded bytes differ"
        hip.hipFree(out)
        print(f"{mb} MB level {level} frames {frames}: ratio {n / len(comp):.2f} (libzstd compress {t_c:.1f} s)  decode {dt * 1e3:.1f} ms = {n / dt / 1e9:.2f} GB/s of output", flush=True)
