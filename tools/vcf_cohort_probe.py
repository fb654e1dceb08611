"""read_vcf on cohort lines with the reference's real schema (formats = LIST(STRUCT(GT))): SAMPLES (default 100) samples a line,
GB (default 1.0) of file in the page cache -> host DataChunks.  ONE_PASS=1: a single all-columns drain (for EXG_TRACE / rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from exon_duckdb_amd import load_library  # noqa: E402
from exon_duckdb_amd.testing import shapes  # noqa: E402

n_samples = int(os.environ.get("SAMPLES", "100"))
n_lines = max(64, 10_000_000 // (n_samples * 4 + 80))
_, block, _ = shapes.vcf_multisample_block(n_lines, n_samples, seed=n_samples)
hdr = shapes.vcf_cohort_header(n_samples)
reps = max(1, int(float(os.environ.get("GB", "1.0")) * 1e9) // len(block))
path = "/tmp/exg_cohort.vcf"
with open(path, "wb") as f:
    f.write(hdr)
    for _ in range(reps):
        f.write(block)
n = len(hdr) + reps * len(block)
lib = load_library()
if os.environ.get("ONE_PASS"):
    bench.reader_chunks(lib, path, "vcf")
    rows, chunks, dt = bench.reader_chunks(lib, path, "vcf")
    print(f"one pass: {dt * 1e3:.1f} ms", flush=True)
else:
    bench.reader_count(lib, path, "vcf")
    _, dt_c = min((bench.reader_count(lib, path, "vcf") for _ in range(3)), key=lambda x: x[1])
    rows, chunks, dt_a, st = bench.timed_reader_chunks(lib, path, "vcf")
    _, _, dt_f, st_f = bench.timed_reader_chunks(lib, path, "vcf", columns=1 << 8)
    print(f"{n / 1e9:.2f} GB, {n_samples} samples a line: COUNT(*) {dt_c * 1e3:.1f} ms; all columns {dt_a * 1e3:.1f} ms = {n / dt_a / 1e9:.2f} GB/s, "
          f"{st['host_vector_bytes'] / dt_a / 1e9:.1f} GB/s of vectors, nested {st['nested_ns'] * 1e-6:.2f} ms = {n / (st['nested_ns'] * 1e-9) / 1e9:.0f} GB/s; "
          f"formats only {dt_f * 1e3:.1f} ms", flush=True)
os.unlink(path)
