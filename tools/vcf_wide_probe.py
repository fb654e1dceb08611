"""read_vcf on wide (multi-sample) lines, the any-shape scan alone: does it matter WHERE the nine column vectors lie relative to
each other?  (a) torch's own allocations (consecutive blocks of one size: the same row of every column is a multiple of 2 MiB
apart), (b) the columns staggered by k x STAGGER bytes.  VW_GB (3), VW_SAMPLES (100), VW_LINES (100000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exon_duckdb_amd import abi, device
from exon_duckdb_amd.testing import shapes
gb = float(os.environ.get("VW_GB", "3"))
ns, nl = int(os.environ.get("VW_SAMPLES", "100")), int(os.environ.get("VW_LINES", "100000"))
hdr, block, e = shapes.vcf_multisample_block(nl, ns, seed=ns)
reps = max(1, int(gb * 1e9) // len(block))
n = len(hdr) + reps * len(block)
d = torch.zeros((n + 15) // 16 * 16 + 64, dtype=torch.uint8, device="cuda")
d[:len(hdr)].copy_(torch.frombuffer(bytearray(hdr), dtype=torch.uint8))
d[len(hdr):n] = torch.frombuffer(bytearray(block), dtype=torch.uint8).cuda().repeat(reps)
rows = nl * reps
scan = device.VcfScan(n, capacity_records=rows + 16)
kw = dict(n_bytes=n, lead=len(hdr), payload_base=bench.BASE, algo=abi.EXG_ALGO_FUSED_FULL)
def run(tag):
    ms, mn = bench.timed_launches(torch, lambda: scan.launch(d, **kw), 6, warm=1)
    r = scan.fetch()
    print(f"{tag:42s} {ms:7.3f} ms = {n / ms / 1e6:7.1f} GB/s  rows {int(r.n_records)} err {r.error_code}", flush=True)
print("column base addresses mod 2 MiB:", [c.data_ptr() % (2 << 20) for c in scan.cols], " strides:", [scan.cols[k + 1].data_ptr() - scan.cols[k].data_ptr() for k in range(8)])
run("torch allocations")
for stagger in (4096, 65536 + 4096, (1 << 20) + 12288, 256):
    cap = rows + 16
    big = torch.empty((9 * (cap + stagger // 16 + 64), 2), dtype=torch.int64, device="cuda")
    per = cap + stagger // 16
    scan.cols = [big[k * per:k * per + cap] for k in range(9)]
    run(f"one block, column k at k x (cap x 16 + {stagger}) B")
    del big
