"""A FASTA file above the one-pinned-block limit (512 MiB) through the reader: whole-file batch uploaded through
pinned windows; COUNT(*), all columns as chunks, and the first / last rows against the oracle's rows of one body."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import table_function
from oracle import pyoracle
body = bytes(pyoracle.synth_fasta(20000))
exp = pyoracle.fasta_parse(body)
K = int(float(os.environ.get("FASTA_GB", "1.5")) * 1e9 / len(body)) + 1
path = "/tmp/exg_big.fasta"
with open(path, "wb") as f:
    for _ in range(K):
        f.write(body)
nb = K * len(body)
con = table_function.connect()
rel = con.table_function("read_fasta", path)
rel.count()
t0 = time.time(); n = rel.count(); dt = time.time() - t0
assert n == K * exp.n_rows, (n, K * exp.n_rows)
print(f"{nb / 1e9:.2f} GB FASTA, {n} records: COUNT(*) {dt:.3f} s = {nb / dt / 1e9:.1f} GB/s", flush=True)
t0 = time.time(); sizes = rel.chunk_sizes(); dt = time.time() - t0
assert sum(sizes) == n
print(f"all columns as chunks: {dt:.3f} s = {nb / dt / 1e9:.1f} GB/s", flush=True)
rows = rel.fetchall(limit=3)
want = [tuple(exp.columns[k].row(i) for k in ("id", "description", "sequence")) for i in range(3)]
assert rows == want, (rows[0][:2], want[0][:2])
print("first rows equal the oracle's")
os.unlink(path)
