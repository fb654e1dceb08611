"""Soak run at real sizes: an 8 GB FASTQ-150 file through both reader boundaries (32 device batches, prefetch, pools),
with per-chunk checks against the generator's closed form; then 2 GB of VCF through new_reader."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from exon_duckdb_amd import device, table_function
from exon_duckdb_amd.arrow import new_reader
from exon_duckdb_amd.table_function import _decode_strings

GB = float(os.environ.get("SOAK_GB", "8"))
n_rec = int(GB * 1e9) // 332
path = "/tmp/exg_soak.fastq"
piece = 3_000_000
with open(path, "wb") as f:
    for k0 in range(0, n_rec, piece):
        k1 = min(n_rec, k0 + piece)
        f.write(device.synth_fastq((k1 - k0) * 332, file_offset=k0 * 332)[: (k1 - k0) * 332].cpu().numpy().tobytes())
print("file written", os.path.getsize(path), flush=True)
con = table_function.connect()
rel = con.table_function("read_fastq", path)
t0 = time.time(); n = rel.count(); dt = time.time() - t0
assert n == n_rec
print(f"count {n} in {dt:.2f} s = {n_rec*332/dt/1e9:.1f} GB/s", flush=True)
# chunks: first and last name of every chunk must be SYN%012d of its row index
t0 = time.time()
seen = [0]
def first_last(ch):
    # (called while the chunk is alive) first and last name of every 257th chunk
    seen[0] += 1
    if seen[0] % 257 != 1:
        return None
    k = int(ch.n_rows)
    names = _decode_strings(ch.vectors[0].contents.data, None, k)
    return names[0], names[-1]
row = 0; n_chunks = 0
for _, k, names in rel._scan([0, 1], decode=first_last):
    if names is not None:
        assert names[0] == b"SYN%012d" % row and names[1] == b"SYN%012d" % (row + k - 1), (row, names[0])
    row += k; n_chunks += 1
dt = time.time() - t0
assert row == n_rec
print(f"chunks {n_chunks} rows {row} in {dt:.2f} s = {n_rec*332/dt/1e9:.1f} GB/s", flush=True)
t0 = time.time(); row = 0
for i, b in enumerate(new_reader(path, "fastq", filters="description='2:N:0:ACGT'")):
    if i % 97 == 0:
        assert b.column(1)[0].as_py() == "2:N:0:ACGT"
    row += b.num_rows
dt = time.time() - t0
assert row == (n_rec + 1) // 4 or row == n_rec // 4, (row, n_rec)
print(f"arrow + filter rows {row} in {dt:.2f} s = {n_rec*332/dt/1e9:.1f} GB/s", flush=True)
os.remove(path)
from oracle import pyoracle
vpath = "/tmp/exg_soak.vcf"
body = bytes(pyoracle.synth_vcf(1_000_000))
hdr = int(pyoracle.vcf_parse(body[:4096], want_string_t=False).extra["header_bytes"])
reps = 40
with open(vpath, "wb") as f:
    f.write(body[:hdr])
    for _ in range(reps):
        f.write(body[hdr:])
t0 = time.time(); rows = 0; dp = 0
for b in new_reader(vpath, "vcf"):
    rows += b.num_rows
dt = time.time() - t0
assert rows == 1_000_000 * reps, rows
print(f"vcf nested rows {rows} in {dt:.2f} s = {os.path.getsize(vpath)/dt/1e9:.1f} GB/s", flush=True)
os.remove(vpath)
print("soak ok")
