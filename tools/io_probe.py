"""COUNT(*) and chunk rates of the plain-file reader under different pread thread counts / slice sizes
(EXG_IO_THREADS, EXG_IO_SLICE_MB are read once per process: run one setting per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import device, table_function
path = "/tmp/exg_bench.fastq"
nb = 332 * 12_000_000
if not os.path.exists(path) or os.path.getsize(path) != nb:
    with open(path, "wb") as f:
        f.write(device.synth_fastq(nb)[:nb].cpu().numpy().tobytes())
con = table_function.connect()
rel = con.table_function("read_fastq", path)
for label, fn in (("count", rel.count), ("chunks", lambda: sum(rel.chunk_sizes()))):
    fn()
    best = 1e9
    for _ in range(3):
        t0 = time.time(); n = fn(); best = min(best, time.time() - t0)
    print(os.environ.get("EXG_IO_THREADS", "8"), os.environ.get("EXG_IO_SLICE_MB", "8"), label, f"{nb/best/1e9:.2f} GB/s", flush=True)
