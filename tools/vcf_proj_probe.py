"""read_vcf of a VCF-8 file in the page cache, projected (chrom, pos, ref) and all columns, through exg_open / exg_next_chunk (the C
drain loop): where a projected batch's time goes (EXG_TRACE=1 prints the stages of every batch).  VP_GB (2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exon_duckdb_amd import device, load_library
lib = load_library()
n_lines = int(float(os.environ.get("VP_GB", "2")) * 1e9 / 48.65)
d, n = device.synth_vcf(n_lines)
p = "/dev/shm/exg_vp.vcf"
bench.write_device_bytes(torch, d, n, p)
del d
try:
    bench.reader_count(lib, p, "vcf")
    for name, cols in (("COUNT(*)", None), ("chrom,pos,ref", 0b1011), ("all columns", 0)):
        best = 1e9
        for _ in range(3):
            if cols is None:
                _, dt = bench.reader_count(lib, p, "vcf")
            else:
                _, _, dt = bench.reader_chunks(lib, p, "vcf", columns=cols)
            best = min(best, dt)
        print(f"{name:14s} {best * 1e3:7.1f} ms = {n / best / 1e9:5.1f} GB/s", flush=True)
finally:
    os.unlink(p)
