"""VCF 8-column scan (BASELINE config 3) alone, for profiling: VCF_GB (5) GB generated in HBM by exg_synth_vcf."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exon_duckdb_amd import abi, device
torch.cuda.set_device(0)
n_lines = int(float(os.environ.get("VCF_GB", "5")) * 1e9 / 48.65)
d_in, n = device.synth_vcf(n_lines)
head = bytes(d_in[:4096].cpu().numpy())
hdr = head.index(b"#CHROM")
hdr += head[hdr:].index(b"\n") + 1
scan = device.VcfScan(n, capacity_records=n_lines + 16)
out = {}
for label, proj in (("all_columns", None), ("chrom_pos_only", {0})):
    for _ in range(2):
        scan.launch(d_in, n_bytes=n, lead=hdr, algo=abi.EXG_ALGO_FUSED, project=proj)
    torch.cuda.synchronize()
    ev = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); scan.launch(d_in, n_bytes=n, lead=hdr, algo=abi.EXG_ALGO_FUSED, project=proj); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)[3]
    r = scan.fetch()
    assert r.error_code == 0 and r.n_records == n_lines
    out[label] = {"bytes": n, "lines": n_lines, "ms": ms, "read_GBps": n / ms / 1e6}
print(json.dumps(out))
