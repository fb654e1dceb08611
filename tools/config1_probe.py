"""BASELINE config 1: SELECT COUNT(*) FROM read_fasta() on a 1 MB FASTA — through the table function (open, upload,
scan, close) and the device scan alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exon_duckdb_amd import device, table_function
from oracle import pyoracle     # writes the input only
data = bytes(pyoracle.synth_fasta(560))[:1_000_000]
data = data[: data.rfind(b"\n>") + 1]
path = "/tmp/exg_cfg1.fasta"
open(path, "wb").write(data)
exp = pyoracle.fasta_parse(data)
con = table_function.connect()
rel = con.table_function("read_fasta", path)
assert rel.count() == exp.n_rows
ts = []
for _ in range(50):
    t0 = time.perf_counter(); n = rel.count(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"{len(data)} B FASTA, {n} records: SELECT COUNT(*) median {ts[25] * 1e3:.2f} ms (min {ts[0] * 1e3:.2f} ms)")
d_in = device.upload(data)
scan = device.FastaScan(len(data), capacity_records=exp.n_rows + 16)
for _ in range(5):
    scan.launch(d_in)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
for a, b in ev:
    a.record(); scan.launch(d_in); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
assert scan.fetch().n_records == exp.n_rows
print(f"device scan alone (six launches): median {ms[100] * 1e3:.1f} us = {len(data) / ms[100] / 1e6:.1f} GB/s")
os.unlink(path)
