"""Is the copy engine a stream's copies run on a property of the STREAM (sticky), and do some pairs of streams share one?  N streams;
for every ordered pair (a: 256 MiB H2D in 8 MiB slices like an upload, b: one 256 MiB D2H at the same time) the aggregate GB/s —
~97 when the two directions overlap, ~57 when they queue on one engine.  Then the same matrix again (is it the same?).  Run on the GPU box."""
import sys, time
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = 256 << 20
d_up, d_dn = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
h_up, h_dn = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream() for _ in range(N)]
sl = 8 << 20


def pair(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[b]):
        h_dn.copy_(d_dn, non_blocking=True)
    with torch.cuda.stream(streams[a]):
        for o in range(0, n, sl):
            d_up[o:o + sl].copy_(h_up[o:o + sl], non_blocking=True)
    torch.cuda.synchronize()
    return 2 * n / (time.perf_counter() - t0) / 1e9


for a in range(N):
    for b in range(N):
        if a != b:
            pair(a, b)   # warm: the streams' queues exist
        break
for rnd in range(2):
    print(f"round {rnd}: rows = the upload's stream, columns = the D2H copy's stream, GB/s in sum")
    for a in range(N):
        print("  " + " ".join(f"{pair(a, b):5.1f}" if a != b else "    -" for b in range(N)), flush=True)
