"""The VCF Arrow stream (new_reader, nested columns) alone: VCF_LINES synthetic lines; EXG_TRACE=1 for the emit stage times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import device
from exon_duckdb_amd.arrow import new_reader
n = int(os.environ.get("VCF_LINES", "4000000"))
t, nb = device.synth_vcf(n)
path = "/tmp/exg_probe.vcf"
open(path, "wb").write(t[:nb].cpu().numpy().tobytes())
for _ in range(3):
    t0 = time.time(); rows = sum(b.num_rows for b in new_reader(path, "vcf")); dt = time.time() - t0
    assert rows == n
    print(f"vcf arrow {rows} {dt:.3f}s {nb/dt/1e9:.2f} GB/s {rows/dt/1e6:.1f} M rows/s", flush=True)
