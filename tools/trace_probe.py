import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd.arrow import new_reader
from oracle import pyoracle
vpath = "/tmp/exg_bench.vcf"
data = bytes(pyoracle.synth_vcf(4_000_000))
open(vpath, "wb").write(data)
for i in range(6):
    t0 = time.time(); rdr = new_reader(vpath, "vcf"); t1 = time.time()
    n = 0
    for b in rdr:
        n += b.num_rows
    t2 = time.time()
    del rdr, b
    t3 = time.time()
    print(f"call {i}: open {t1-t0:.3f} read {t2-t1:.3f} release {t3-t2:.3f}", flush=True)
