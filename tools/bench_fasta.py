"""FASTA scan throughput (GPU box): ~1 GB of 60-column wrapped records generated in HBM by exg_synth_fasta."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exon_duckdb_amd import device
torch.cuda.set_device(0)
n_rec = int(os.environ.get("FASTA_RECORDS", "600000"))
d_in, n = device.synth_fasta(n_rec)
scan = device.FastaScan(n, capacity_records=n_rec + 16)
for _ in range(2):
    scan.launch(d_in)
torch.cuda.synchronize()
ev = []
for _ in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); scan.launch(d_in); b.record(); ev.append((a, b))
torch.cuda.synchronize()
ms = sorted(x.elapsed_time(y) for x, y in ev)[3]
r = scan.fetch()
assert r.error_code == 0 and r.n_records == n_rec, (r.error_code, r.n_records)
print(json.dumps({"bytes": n, "records": n_rec, "ms": ms, "GBps": n / ms / 1e6}))
