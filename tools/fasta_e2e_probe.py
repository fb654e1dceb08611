"""read_fasta end to end on a synthetic FASTA file (GB, default 2) in the page cache: COUNT(*) and the all-columns drain, best of 3;
EXG_TRACE=2 shows a pass's timeline (the last drain).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from exon_duckdb_amd import device, load_library  # noqa: E402

gb = float(os.environ.get("GB", "2"))
p = os.path.join("/dev/shm", "fa_probe.fasta")
if os.environ.get("KIND") == "genome":
    # chromosome-sized records (REC_MB of bases each, lines of 60): what a reference genome looks like — a record is many batches long
    import numpy as np
    rec_mb = float(os.environ.get("REC_MB", "125"))
    rng = np.random.default_rng(5)
    n = 0
    with open(p, "wb") as f:
        k = 0
        while n < gb * 1e9:
            lines = int(rec_mb * 1e6) // 60 + int(rng.integers(0, 1000))
            body = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (lines, 61), dtype=np.uint8)]
            body[:, 60] = 10
            hdr = b">chr%d synthetic chromosome %d\n" % (k + 1, k + 1)
            f.write(hdr)
            f.write(body.tobytes())
            n += len(hdr) + body.size
            k += 1
    n_rec = k
else:
    n_rec = int(gb * 1e9 / 1712)
    d, n = device.synth_fasta(n_rec)
    bench.write_device_bytes(torch, d, n, p)
    del d
if os.environ.get("BGZF"):   # the same file bgzip'd (x.fasta.gz): the decoded segments are scanned from an aligned device copy
    from exon_duckdb_amd.testing.bgzf import bgzip
    nz = bgzip(p, p + ".gz")
    os.unlink(p)
    p = p + ".gz"
    print(f"bgzip: {n/1e9:.2f} GB in {nz/1e9:.2f} GB", flush=True)
lib = load_library()
bench.reader_count(lib, p, "fasta")
t_c = min(bench.reader_count(lib, p, "fasta")[1] for _ in range(3))
t_a = min(bench.reader_chunks(lib, p, "fasta")[2] for _ in range(3))
print(f"read_fasta {n/1e9:.2f} GB, {n_rec} records: COUNT(*) {t_c*1e3:.1f} ms = {n/t_c/1e9:.1f} GB/s, all columns {t_a*1e3:.1f} ms = {n/t_a/1e9:.1f} GB/s", flush=True)
if os.environ.get("ARROW"):   # the reference's own boundary: new_reader -> Arrow C stream, batches pulled by pyarrow
    import time
    from exon_duckdb_amd.arrow import new_reader
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        rows = sum(b.num_rows for b in new_reader(p, "fasta"))
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    assert rows == n_rec
    print(f"new_reader(fasta) -> Arrow record batches (pyarrow loop): {best*1e3:.1f} ms = {n/best/1e9:.1f} GB/s", flush=True)
os.unlink(p)
