#!/usr/bin/env python3
"""read_fastq end to end (file in the page cache -> COUNT(*) / host DataChunks through the C drain loop), best of 3:
python tools/e2e_probe.py [GB] — what bench.py's end_to_end leg measures; EXG_ZERO_BOUNCE=0 takes the bounce path"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exon_duckdb_amd import device, load_library
lib = load_library()
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gb * 1e9) // 332 * 332
d = tempfile.mkdtemp(prefix="exg_e2e_", dir="/dev/shm")
p = os.path.join(d, "x.fastq")
with open(p, "wb") as f:
    step = (1 << 30) // 332 * 332
    for o in range(0, n, step):
        m = min(step, n - o)
        f.write(device.synth_fastq(m, file_offset=o)[:m].cpu().numpy().tobytes())
torch.cuda.empty_cache()
bench.reader_count(lib, p, "fastq")
rows, dt_c = min((bench.reader_count(lib, p, "fastq") for _ in range(3)), key=lambda x: x[1])
r2, chunks, dt_r = min((bench.reader_chunks(lib, p, "fastq") for _ in range(3)), key=lambda x: x[2])
assert rows == r2 == n // 332
print(f"COUNT(*) {n / dt_c / 1e9:.1f} GB/s  chunks {n / dt_r / 1e9:.1f} GB/s")
os.unlink(p); os.rmdir(d)
