"""read_vcf end to end on a generated VCF-8 file in the page cache (round 6: the nested chain rebuilt): COUNT(*), all nine
columns, chrom/pos/ref projected — best of 3 each — then one pass with EXG_TRACE for the stages.  VCF_LINES (default 42.7 M = 2.08 GB)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from exon_duckdb_amd import device, load_library  # noqa: E402

n_lines = int(os.environ.get("VCF_LINES", "42700000"))
path = "/tmp/exg_nested.vcf"
t, n = device.synth_vcf(n_lines)
bench.write_device_bytes(torch, t, n, path)
del t
torch.cuda.empty_cache()
lib = load_library()
if os.environ.get("ONE_PASS"):
    # PROBE_COLUMNS=2: only `pos` travels — the nested columns are still made (and validated) on the device, so a kernel trace shows
    # their kernels without the profiler's copy kernels beside them (under rocprofv3 the big D2H copies of this process run as blit
    # kernels and stretch every kernel that runs at the same time)
    rows, chunks, dt = bench.reader_chunks(lib, path, "vcf", columns=int(os.environ.get("PROBE_COLUMNS", "0")))
    print(f"one pass: {dt * 1e3:.1f} ms", flush=True)
    os.unlink(path)
    sys.exit(0)
bench.reader_count(lib, path, "vcf")
_, dt_c = min((bench.reader_count(lib, path, "vcf") for _ in range(3)), key=lambda x: x[1])
rows, chunks, dt_a = min((bench.reader_chunks(lib, path, "vcf") for _ in range(3)), key=lambda x: x[2])
_, _, dt_p = min((bench.reader_chunks(lib, path, "vcf", columns=0b1011) for _ in range(3)), key=lambda x: x[2])
assert rows == n_lines
print(f"{n / 1e9:.2f} GB: COUNT(*) {dt_c * 1e3:.1f} ms = {n / dt_c / 1e9:.1f} GB/s; all columns {dt_a * 1e3:.1f} ms = {n / dt_a / 1e9:.2f} GB/s; "
      f"chrom,pos,ref {dt_p * 1e3:.1f} ms = {n / dt_p / 1e9:.1f} GB/s", flush=True)
os.environ["EXG_TRACE"] = "1"
bench.reader_chunks(lib, path, "vcf")
os.unlink(path)
