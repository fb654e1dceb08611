#!/usr/bin/env python3
"""page cache -> HBM without the bounce copy (run on the GPU box): python tools/zero_bounce_probe.py [GB]
the bounce path (N readers x adaptive pread threads -> pinned block -> H2D) against hipHostRegister of the file mapping's
windows and against O_DIRECT reads into pinned blocks, at 1 / 2 / 4 / 8 readers on this box's one GPU"""
import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exon_duckdb_amd import device, load_test_library
tl = load_test_library()
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gb * 1e9) // 332 * 332
d = tempfile.mkdtemp(prefix="exg_zb_", dir="/dev/shm")
p = os.path.join(d, "x.fastq")
with open(p, "wb") as f:
    step = (1 << 30) // 332 * 332
    for o in range(0, n, step):
        m = min(step, n - o)
        f.write(device.synth_fastq(m, file_offset=o)[:m].cpu().numpy().tobytes())
cores = bench.effective_cores()
print("usable cores", cores)
for k in (1, 2, 4, 8):
    t = min(8, max(2, cores // k))
    a = tl.exon_tf_host_pipeline_probe(p.encode(), k, t, 1, 0, 1.0)
    ms = C.c_double(0)
    b = tl.exon_tf_host_zero_bounce_probe(p.encode(), k, 1, 0, 1.5, C.byref(ms))
    c = tl.exon_tf_host_zero_bounce_probe(p.encode(), k, 2, 0, 1.0, None)
    ms3, ms4 = C.c_double(0), C.c_double(0)
    b3 = tl.exon_tf_host_zero_bounce_probe(p.encode(), k, 3, 0, 1.5, C.byref(ms3))
    b4 = tl.exon_tf_host_zero_bounce_probe(p.encode(), k, 4, 0, 1.5, C.byref(ms4))
    print(f"{k} readers: bounce ({t} threads each) + h2d {a/1e9:.1f} GB/s | hipHostRegister: windows of one warm mapping {b/1e9:.1f} GB/s ({ms.value:.1f} ms per "
          f"register+unregister of 256 MiB), a fresh mapping per window {b3/1e9:.1f} ({ms3.value:.1f} ms), + entries made by 4 threads first {b4/1e9:.1f} "
          f"({ms4.value:.1f} ms) | O_DIRECT {c/1e9 if c > 0 else c:.1f}")
os.unlink(p); os.rmdir(d)
