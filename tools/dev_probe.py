"""Dev probes (GPU box): MALL residency of a re-read stream; cost split of the fused kernel."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from exon_duckdb_amd import abi, device, load_library

lib = load_library()
torch.cuda.set_device(0)


def time_ms(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2]


out = {}
n = 4 * 10**9
d_in = device.synth_fastq(n)
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
# 1. streaming count of the same X bytes, back to back: does the second pass come from the Infinity Cache?
for mb in (4000,):
    x = mb * 10**6 // 16 * 16
    ms = time_ms(lambda: lib.exg_count_newlines(C.c_void_p(d_in.data_ptr()), 0, x, C.c_void_p(cnt.data_ptr()),
                                                device.stream_ptr()), reps=20, warm=3)
    out[f"count_rereads_{mb}MB_GBps"] = x / ms / 1e6
print(json.dumps(out, indent=1))

# 2. fused kernel with dev modes (flags bits 8..): see exg_fastq_fused.hip
scan = device.FastqScan(n, capacity_records=n // 332 + 16)
for mode in (0, 1, 2, 3, 4):
    fl = abi.EXG_F_BOF | abi.EXG_F_EOF | (mode << 8)
    ms = time_ms(lambda: scan.launch(d_in, n_bytes=n, flags=fl, algo=abi.EXG_ALGO_FUSED), reps=10, warm=2)
    r = scan.fetch()
    print(f"fused dev_mode={mode}: {ms:.3f} ms  {n / ms / 1e6:.0f} GB/s  n_records={r.n_records} err={r.error_code} flags={r.flags}")

