"""Which engine moves a pinned-host -> device copy (SDMA: ~53 GB/s; blit kernel: ~49 GB/s and 256 workgroups on the CUs):
torch's copy against hipMemcpyAsync on a non-blocking stream, from one thread and from eight."""
import ctypes as C, os, sys, threading, time
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
n, sl = 1 << 30, 8 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h_t = torch.empty(n, dtype=torch.uint8).pin_memory()
hp = C.c_void_p()
assert hip.hipHostMalloc(C.byref(hp), n, 0) == 0
streams = {}
for name, flags in (("default-flags", 0), ("non-blocking", 1)):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), flags) == 0
    streams[name] = s


def run(label, src, stream, threads):
    def work(k):
        hip.hipSetDevice(0)
        for o in range(k * sl, n, sl * threads):
            assert hip.hipMemcpyAsync(d.data_ptr() + o, src + o, sl, 1, stream) == 0
    best = 0
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        [t.start() for t in th]; [t.join() for t in th]
        hip.hipStreamSynchronize(stream)
        best = max(best, n / (time.perf_counter() - t0) / 1e9)
    print(f"{label}: {best:.1f} GB/s", flush=True)


s = torch.cuda.Stream()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for o in range(0, n, sl):
            d[o:o + sl].copy_(h_t[o:o + sl], non_blocking=True)
    s.synchronize(); dt = time.perf_counter() - t0
print(f"torch copy_: {n / dt / 1e9:.1f} GB/s", flush=True)
for sname, st in streams.items():
    run(f"hipMemcpyAsync, torch pinned, {sname} stream, 1 thread", h_t.data_ptr(), st, 1)
    run(f"hipMemcpyAsync, hipHostMalloc, {sname} stream, 1 thread", hp.value, st, 1)
    run(f"hipMemcpyAsync, hipHostMalloc, {sname} stream, 8 threads", hp.value, st, 8)
