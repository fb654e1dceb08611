"""All four columns of a BGZF FASTQ file as DataChunks, drained by a C loop (no interpreter between the chunks), against
COUNT(*): GZC_GB (4) GB of compressed bytes; EXG_TRACE=1 shows the stages of every batch."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exon_duckdb_amd import device, load_library
lib = load_library()
gb = float(os.environ.get("GZC_GB", "4"))
n_rec = 100_000
raw = device.synth_fastq(332 * n_rec)[: 332 * n_rec].cpu().numpy().tobytes()
parts = []
for i in range(0, len(raw), 65280):
    chunk = raw[i:i + 65280]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    d = co.compress(chunk) + co.flush()
    parts.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1)
                 + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
block = b"".join(parts)
K = max(1, int(gb * 1e9 / len(block)))
path = "/dev/shm/exg_gzc.fastq.gz"
with open(path, "wb") as f:
    for _ in range(K):
        f.write(block)
infl = K * len(raw)
try:
    for rep in range(2):
        n, dt = bench.reader_count(lib, path, "fastq", (0, 1), 0)
        print(f"COUNT(*): {dt:.3f} s = {infl / dt / 1e9:.1f} GB/s of FASTQ", flush=True)
    for rep in range(2):
        rows, chunks, dt = bench.reader_chunks(lib, path, "fastq")
        assert rows == K * n_rec
        print(f"all columns: {dt:.3f} s = {infl / dt / 1e9:.1f} GB/s of FASTQ ({chunks} chunks)", flush=True)
finally:
    os.unlink(path)
