"""read_fastq end to end on the 4 GB FASTQ-150 file of bench.py's `end_to_end` leg (page cache -> host DataChunks): COUNT(*) and the
all-columns drain, RUNS (7) times each, every time printed (the leg's run-to-run spread is the question); knobs from the
environment: EXG_IO_THREADS, EXG_IO_SLICE_MB, EXG_NO_RAMP, EXG_NO_RUN_AHEAD.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from exon_duckdb_amd import device, load_library  # noqa: E402

n = int(float(os.environ.get("GB", "4")) * 1e9) // 332 * 332
p = os.environ.get("FQ_FILE") or "/dev/shm/fq_probe.fastq"
if not os.path.exists(p):
    bench.write_device_bytes(torch, device.synth_fastq(n)[:n], n, p)
if os.environ.get("BGZF") and not p.endswith(".gz"):   # the same file bgzip'd (x.fastq.gz)
    from exon_duckdb_amd.testing.bgzf import bgzip
    if not os.path.exists(p + ".gz"):
        bgzip(p, p + ".gz")
    if not os.environ.get("FQ_FILE"):
        os.unlink(p)
    p = p + ".gz"
lib = load_library()
runs = int(os.environ.get("RUNS", "7"))
bench.reader_count(lib, p, "fastq")
tc = [bench.reader_count(lib, p, "fastq")[1] * 1e3 for _ in range(runs)]
ta = [bench.reader_chunks(lib, p, "fastq")[2] * 1e3 for _ in range(runs)]
print(f"{os.environ.get('LABEL', '')} COUNT(*) " + " ".join(f"{x:.1f}" for x in tc) + " | all columns " + " ".join(f"{x:.1f}" for x in ta) +
      f" | best {min(ta):.1f} ms = {n / min(ta) / 1e6:.1f} GB/s, median {sorted(ta)[runs // 2]:.1f}", flush=True)
if os.environ.get("FILTERS"):   # the `filters` predicate evaluated on the device (exg_open_args.filters: what filter pushdown hands down)
    f = os.environ["FILTERS"]
    rows_c, _ = bench.reader_count(lib, p, "fastq", filters=f)
    tfc = [bench.reader_count(lib, p, "fastq", filters=f)[1] * 1e3 for _ in range(3)]
    out = [bench.reader_chunks(lib, p, "fastq", filters=f) for _ in range(3)]
    print(f"filters {f!r}: {rows_c} of {n // 332} rows; COUNT(*) " + " ".join(f"{x:.1f}" for x in tfc) + " ms | chunks " +
          " ".join(f"{x[2] * 1e3:.1f}" for x in out) + f" ms ({out[0][0]} rows in {out[0][1]} chunks)", flush=True)
if os.environ.get("ARROW"):   # the reference's own boundary: new_reader -> Arrow C stream, record batches pulled and released by a C loop
    import ctypes as C
    from exon_duckdb_amd import load_test_library
    tl = load_test_library()
    def drain():
        import time
        rows, batches, dg = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        err = C.create_string_buffer(512)
        t0 = time.perf_counter()
        rc = tl.exon_tf_drain_arrow_fastq(p.encode(), None, None, 0, C.byref(rows), C.byref(batches), C.byref(dg), err, 512)
        dt = time.perf_counter() - t0
        assert rc == 0 and rows.value == n // 332, err.value
        return dt * 1e3
    drain()
    tr = [drain() for _ in range(runs)]
    print(f"{os.environ.get('LABEL', '')} new_reader -> Arrow record batches " + " ".join(f"{x:.1f}" for x in tr) + f" | best {min(tr):.1f} ms = {n / min(tr) / 1e6:.1f} GB/s", flush=True)
if not os.environ.get("FQ_FILE"):
    os.unlink(p)
