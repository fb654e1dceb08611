"""COUNT(*) through the reader on a .zst of FASTQ-150 (level 3; ZST_CHECK=0: no Content_Checksum) for several round sizes:
ZST_GB (2) GB of content in ONE frame, or in frames of ZST_FRAME_MB MiB of content each (what pzstd and the seekable format
write); run on the GPU box."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from exon_duckdb_amd import device
from exon_duckdb_amd.reader import ShardReader
from zstd_util import compress

gb = float(os.environ.get("ZST_GB", "2"))
n = int(gb * 1e9) // 332 * 332
keep = os.environ.get("ZST_FILE")   # a path: the file is built once and kept (several runs of the probe under different switches in one box)
def build():
    if os.environ.get("ZST_KIND") == "genome":
        # reads sampled from a small genome (deep coverage: long matches far back, what a resequencing run looks like to an LZ coder)
        # with qualities in runs — 332-byte records like exg_synth_fastq's, so that the row count is the same closed form
        import numpy as np
        rng = np.random.default_rng(3)
        genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(float(os.environ.get("ZST_GENOME_MB", "2")) * 1e6))
        n_rec = n // 332
        rec = np.empty((n_rec, 332), dtype=np.uint8)
        rec[:] = np.frombuffer(b"@" + b"G" * 26 + b"\n" + b"A" * 150 + b"\n+\n" + b"I" * 150 + b"\n", dtype=np.uint8)
        ids = np.char.zfill(np.arange(n_rec).astype("U12"), 12)
        rec[:, 1:13] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(n_rec, 12)
        rec[:, 13:27] = np.frombuffer(b" 1:N:0:ACGTACG", dtype=np.uint8)
        pos = rng.integers(0, len(genome) - 150, n_rec)
        rec[:, 28:178] = genome[pos[:, None] + np.arange(150)[None, :]]
        q = np.repeat(rng.integers(35, 74, (n_rec, 15), dtype=np.uint8), 10, axis=1)
        rec[:, 181:331] = q
        data = rec.tobytes()
        del rec, q, pos, ids
    else:
        data = device.synth_fastq(n)[:n].cpu().numpy().tobytes()
    t0 = time.time()
    frame_mb = float(os.environ.get("ZST_FRAME_MB", "0"))   # (fractions: ZST_FRAME_MB=0.0625 = frames of 64 KiB)
    level, wlog = int(os.environ.get("ZST_LEVEL", "3")), int(os.environ.get("ZST_WLOG", "0"))
    check = os.environ.get("ZST_CHECK", "1") == "1"
    if frame_mb:
        step = int(frame_mb * (1 << 20)) // 332 * 332
        comp = b"".join(compress(data[o:o + step], level, check, window_log=wlog) for o in range(0, n, step))
    else:
        comp = compress(data, level, check, window_log=wlog)
    print(f"compressed {n/1e9:.2f} GB -> {len(comp)/1e9:.2f} GB in {time.time()-t0:.1f} s", flush=True)
    d = tempfile.mkdtemp(dir="/dev/shm") if not keep else os.path.dirname(keep)
    p = keep or os.path.join(d, "x.fastq.zst")
    open(p, "wb").write(comp)
    return d, p
if keep and os.path.exists(keep):
    d, p = os.path.dirname(keep), keep
else:
    d, p = build()
for batch in os.environ.get("ZST_BATCHES", "0,536870912,1073741824,4294967296").split(","):
    best = None
    for _ in range(3):
        r = ShardReader(p, "fastq", device_batch_bytes=int(batch))
        t0 = time.perf_counter()
        rows = r.count()
        dt = time.perf_counter() - t0
        if os.environ.get("EXG_TRACE"):
            print(f"[probe] count() returned after {dt*1e3:.1f} ms", file=sys.stderr, flush=True)
        st = r.stats()
        r.close()
        assert rows == n // 332
        best = dt if best is None or dt < best else best
    print(f"device_batch_bytes={int(batch) >> 20} MiB: {best*1e3:.1f} ms = {n/best/1e9:.1f} GB/s of FASTQ ({st['decoded_segments']} segments, peak {st['device_bytes_peak']>>20} MiB)", flush=True)
if os.environ.get("ZST_CHUNKS"):
    # every column as DataChunks (the C drain loop of bench.reader_chunks), best of three
    import bench
    from exon_duckdb_amd import load_library
    lib = load_library()
    rows, chunks, dt = min((bench.reader_chunks(lib, p, "fastq") for _ in range(3)), key=lambda x: x[2])
    assert rows == n // 332
    print(f"all columns as DataChunks: {dt*1e3:.1f} ms = {n/dt/1e9:.1f} GB/s of FASTQ ({chunks} chunks)", flush=True)
if not keep:
    os.unlink(p); os.rmdir(d)
