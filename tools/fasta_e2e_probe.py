"""read_fasta end to end on a synthetic FASTA file (GB, default 2) in the page cache: COUNT(*) and the all-columns drain, best of 3;
EXG_TRACE=2 shows a pass's timeline (the last drain).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from exon_duckdb_amd import device, load_library  # noqa: E402

gb = float(os.environ.get("GB", "2"))
n_rec = int(gb * 1e9 / 1712)
d, n = device.synth_fasta(n_rec)
p = os.path.join("/dev/shm", "fa_probe.fasta")
bench.write_device_bytes(torch, d, n, p)
del d
lib = load_library()
bench.reader_count(lib, p, "fasta")
t_c = min(bench.reader_count(lib, p, "fasta")[1] for _ in range(3))
t_a = min(bench.reader_chunks(lib, p, "fasta")[2] for _ in range(3))
print(f"read_fasta {n/1e9:.2f} GB: COUNT(*) {t_c*1e3:.1f} ms = {n/t_c/1e9:.1f} GB/s, all columns {t_a*1e3:.1f} ms = {n/t_a/1e9:.1f} GB/s", flush=True)
os.unlink(p)
