"""Latency of tiny inputs (the reference's fixtures) and the quality_score_string_to_list device op."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exon_duckdb_amd import device, table_function
from exon_duckdb_amd.arrow import new_reader
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
con = table_function.connect()
for fn, name in (("read_fastq", "test.fastq"), ("read_fasta", "test.fasta"), ("read_vcf", "vcf/index.vcf"), ("read_fastq", "test.fastq.gz"),
                 ("read_vcf", "vcf/index.vcf.gz")):
    path = os.path.join(G, name)
    rel = con.table_function(fn, path)
    rel.fetchall()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); rows = rel.fetchall(); ts.append(time.perf_counter() - t0)
    ts.sort()
    t1 = []
    fmt = {"read_fastq": "fastq", "read_fasta": "fasta", "read_vcf": "vcf"}[fn]
    for _ in range(20):
        t0 = time.perf_counter(); n = sum(b.num_rows for b in new_reader(path, fmt)); t1.append(time.perf_counter() - t0)
    t1.sort()
    print(f"{fn}('{name}'): {len(rows)} rows, chunks median {ts[10] * 1e3:.2f} ms (min {ts[0] * 1e3:.2f}), new_reader median {t1[10] * 1e3:.2f} ms", flush=True)

# quality_score_string_to_list straight behind the scan, in HBM
n_rec = 12_000_000
nb = 332 * n_rec
d_in = device.synth_fastq(nb)
scan = device.FastqScan(nb, capacity_records=n_rec + 16)
scan.launch(d_in, payload_base=0x7F0000000000)
assert scan.fetch().n_records == n_rec
torch.cuda.synchronize()
device.quality_score_string_to_list(scan.cols[3], n_rec, d_in, 0x7F0000000000, values_capacity=150 * n_rec)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, v, t = device.quality_score_string_to_list(scan.cols[3], n_rec, d_in, 0x7F0000000000, values_capacity=150 * n_rec)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
assert t == 150 * n_rec
print(f"quality_score_string_to_list: {n_rec} rows, {t} values in {best * 1e3:.2f} ms (incl. allocation) = {n_rec / best / 1e9:.2f} G rows/s, "
      f"{(150 + 600 + 16) * n_rec / best / 1e12:.2f} TB/s of HBM traffic", flush=True)
