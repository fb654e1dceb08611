"""Do hipMemcpyAsync D2H / H2D copies between HBM and pinned memory run on SDMA engines or as blit kernels on this box, and what
does a copy in flight do to a small kernel on another stream?  (round 6: the kernel trace of read_vcf shows __amd_rocclr_copyBuffer
for every large copy, and kernels on other queues ending only when a copy ends.)"""
import os, sys, time
import torch
n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
x = torch.zeros(1 << 20, dtype=torch.float32, device="cuda")
s_copy, s_k = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for label, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
    for _ in range(2):
        with torch.cuda.stream(s_copy):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s_copy):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # a 4 MB elementwise kernel alone, then while a copy is in flight
    def small():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_k):
            e0.record()
            x.add_(1.0)
            e1.record()
        return e0, e1
    e0, e1 = small(); torch.cuda.synchronize(); alone = e0.elapsed_time(e1)
    with torch.cuda.stream(s_copy):
        fn()
    time.sleep(0.002)
    e0, e1 = small(); torch.cuda.synchronize(); busy = e0.elapsed_time(e1)
    print(f"{label}: {n / dt / 1e9:.1f} GB/s; small kernel alone {alone * 1e3:.0f} us, beside the copy {busy * 1e3:.0f} us", flush=True)
# both directions at once (the reader's steady state: the next batch's upload beside this batch's columns going back)
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
s_copy2 = torch.cuda.Stream()
torch.cuda.synchronize()
for order in ("H2D first", "D2H first", "two D2H"):
    t0 = time.perf_counter()
    if order == "H2D first":
        with torch.cuda.stream(s_copy2): d2.copy_(h2, non_blocking=True)
        with torch.cuda.stream(s_copy): h.copy_(d, non_blocking=True)
    elif order == "D2H first":
        with torch.cuda.stream(s_copy): h.copy_(d, non_blocking=True)
        with torch.cuda.stream(s_copy2): d2.copy_(h2, non_blocking=True)
    else:
        with torch.cuda.stream(s_copy): h.copy_(d, non_blocking=True)
        with torch.cuda.stream(s_copy2): h2.copy_(d2, non_blocking=True)
    time.sleep(0.002)
    e0, e1 = small()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{order}: 2 x 1 GiB in {dt * 1e3:.1f} ms = {2 * n / dt / 1e9:.1f} GB/s in sum; small kernel beside them {e0.elapsed_time(e1) * 1e3:.0f} us", flush=True)
