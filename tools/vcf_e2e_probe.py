"""End-to-end (PCIe-inclusive) rates of read_vcf on a generated VCF-8 file in the page cache: COUNT(*), the flat columns as
chunks (exg_open / exg_next_chunk), one projected column.  VCF_LINES (default 40 M = 1.95 GB)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exon_duckdb_amd import device, table_function

n_lines = int(os.environ.get("VCF_LINES", "40000000"))
path = "/tmp/exg_e2e.vcf"
t, n = device.synth_vcf(n_lines)
with open(path, "wb") as f:
    step = 1 << 30
    for o in range(0, n, step):
        f.write(t[o:min(n, o + step)].cpu().numpy().tobytes())
del t
con = table_function.connect()
rel = con.table_function("read_vcf", path)
for label, fn in (("count", rel.count), ("chunks (all columns)", lambda: sum(rel.chunk_sizes())),
                  ("chunks (flat columns: chrom pos id ref qual)", lambda: sum(rel.chunk_sizes(["chrom", "pos", "id", "ref", "qual"]))),
                  ("chunks (pos only)", lambda: sum(rel.chunk_sizes(["pos"])))):
    try:
        fn()
        t0 = time.perf_counter(); rows = fn(); dt = time.perf_counter() - t0
        assert rows == n_lines, (rows, n_lines)
        print(f"{label}: {dt:.3f} s = {n / dt / 1e9:.1f} GB/s of VCF, {rows / dt / 1e6:.0f} M rows/s", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{label}: {type(e).__name__}: {e}", flush=True)
# the library's own loop (exg_drain_chunks: every column of the reader's schema, no Python between the chunks)
import bench  # noqa: E402
from exon_duckdb_amd import load_library  # noqa: E402
lib = load_library()
for rep in range(2):
    rows, chunks, dt = bench.reader_chunks(lib, path, "vcf")
    assert rows == n_lines
print(f"exg_drain_chunks (all columns, C loop): {dt:.3f} s = {n / dt / 1e9:.1f} GB/s of VCF, {rows / dt / 1e6:.0f} M rows/s, {chunks} chunks", flush=True)
os.unlink(path)
