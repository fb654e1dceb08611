"""BASELINE config 4 through the reader: FASTQ-150 in 65 280-byte BGZF members, GZ_SOAK_GB of compressed bytes
(the member block of 100 000 records is compressed once and repeated)."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import device, table_function
from exon_duckdb_amd.arrow import new_reader

gb = float(os.environ.get("GZ_SOAK_GB", "4"))
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
path = "/tmp/exg_soak.fastq.gz"
with open(path, "wb") as f:
    for _ in range(K):
        f.write(block)
comp, infl = K * len(block), K * len(raw)
print(f"{comp / 1e9:.2f} GB compressed, {infl / 1e9:.2f} GB inflated, {K * n_rec} records", flush=True)
con = table_function.connect()
rel = con.table_function("read_fastq", path)
for rep in range(2):
    t0 = time.time(); n = rel.count(); dt = time.time() - t0
    assert n == K * n_rec
    print(f"COUNT(*): {dt:.3f} s = {comp / dt / 1e9:.1f} GB/s compressed, {infl / dt / 1e9:.1f} GB/s of FASTQ, {n / dt / 1e6:.0f} M records/s", flush=True)
if os.environ.get("GZ_SOAK_CHUNKS", "1") == "1":
    t0 = time.time(); rows = sum(rel.chunk_sizes())
    dt = time.time() - t0
    assert rows == K * n_rec
    print(f"chunks: {dt:.3f} s = {infl / dt / 1e9:.1f} GB/s of FASTQ", flush=True)
    t0 = time.time(); rows = 0
    for b in new_reader(path, "fastq"):
        rows += b.num_rows
    dt = time.time() - t0
    assert rows == K * n_rec
    print(f"arrow: {dt:.3f} s = {infl / dt / 1e9:.1f} GB/s of FASTQ", flush=True)
os.unlink(path)
