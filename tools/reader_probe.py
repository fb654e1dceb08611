import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from exon_duckdb_amd import device, table_function
path = "/tmp/exg_bench.fastq"
nb = 332 * 12_000_000
with open(path, "wb") as f:
    f.write(device.synth_fastq(nb)[:nb].cpu().numpy().tobytes())
con = table_function.connect()
rel = con.table_function("read_fastq", path)
for label, fn in (("count", rel.count), ("chunks", lambda: sum(rel.chunk_sizes()))):
    fn()
    for _ in range(2):
        t0 = time.time(); n = fn(); dt = time.time() - t0
        print(label, n, f"{dt:.3f}s {nb/dt/1e9:.2f} GB/s {n/dt/1e6:.1f} M rec/s")
os.remove(path)
