"""VCF 8-column scan (BASELINE config 3) alone, for profiling: 5 GB built in HBM from one synthetic body."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exon_duckdb_amd import abi, device
from oracle import pyoracle
torch.cuda.set_device(0)
L = 400_000
body = pyoracle.synth_vcf(L)
hdr = int(pyoracle.vcf_parse(bytes(body[:4096]), want_string_t=False).extra["header_bytes"])
blen = len(body) - hdr
T = int(float(os.environ.get("VCF_GB", "5")) * 1e9) // blen
n = hdr + T * blen
d_body = torch.frombuffer(bytearray(bytes(body)), dtype=torch.uint8).cuda()
d_in = torch.zeros(n + 80, dtype=torch.uint8, device="cuda")
d_in[:hdr] = d_body[:hdr]
d_in[hdr:n].view(T, blen)[:] = d_body[hdr:]
scan = device.VcfScan(n, capacity_records=T * L + 16)
out = {}
for label, proj in (("all_columns", None), ("chrom_pos_only", {0})):
    for _ in range(2):
        scan.launch(d_in, lead=hdr, algo=abi.EXG_ALGO_AUTO, project=proj)
    torch.cuda.synchronize()
    ev = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); scan.launch(d_in, lead=hdr, algo=abi.EXG_ALGO_AUTO, project=proj); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)[2]
    r = scan.fetch()
    assert r.error_code == 0 and r.n_records == T * L
    out[label] = {"bytes": n, "lines": T * L, "ms": ms, "read_GBps": n / ms / 1e6}
print(json.dumps(out, indent=1))
