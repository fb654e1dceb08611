for t in 4 6 8; do EXG_IO_THREADS=$t timeout 200 python tools/io_probe.py 2>&1 | grep GB/s; done
for sl in 16 32; do EXG_IO_THREADS=8 EXG_IO_SLICE_MB=$sl timeout 200 python tools/io_probe.py 2>&1 | grep GB/s; done
python - <<'PY'
import torch, time
h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for _ in range(2): d.copy_(h, non_blocking=True); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"pinned H2D {5 * (1 << 30) / dt / 1e9:.1f} GB/s")
t0 = time.time()
for _ in range(5): h.copy_(d, non_blocking=True)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"pinned D2H {5 * (1 << 30) / dt / 1e9:.1f} GB/s")
PY
