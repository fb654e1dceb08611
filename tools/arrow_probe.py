import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exon_duckdb_amd import device
from exon_duckdb_amd.arrow import new_reader
path = "/tmp/exg_bench.fastq"
nb = 332 * 12_000_000
with open(path, "wb") as f:
    f.write(device.synth_fastq(nb)[:nb].cpu().numpy().tobytes())
for i in range(3):
    t0 = time.time(); n = sum(b.num_rows for b in new_reader(path, "fastq")); dt = time.time() - t0
    print(f"arrow {n} {dt:.3f}s {nb/dt/1e9:.2f} GB/s", flush=True)
