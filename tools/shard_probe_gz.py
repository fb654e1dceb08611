"""BGZF FASTQ read as 8 shards by member ranges (one after the other on one GPU): each shard uploads and inflates only
its own members (+ ~1 MiB of members in front)."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import device
from exon_duckdb_amd.reader import ShardReader
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
K = max(1, int(float(os.environ.get("GZ_SOAK_GB", "4")) * 1e9 / len(block)))
path = "/tmp/exg_shard.fastq.gz"
with open(path, "wb") as f:
    for _ in range(K):
        f.write(block)
comp, infl, total_rec = K * len(block), K * len(raw), K * n_rec
ShardReader(path, "fastq").count()
t0 = time.time(); whole = ShardReader(path, "fastq").count(); t_whole = time.time() - t0
tot, t_all = 0, 0.0
for i in range(8):
    t0 = time.time(); c = ShardReader(path, "fastq", shard_index=i, shard_count=8).count(); dt = time.time() - t0
    tot += c; t_all += dt
    print(f"shard {i}/8: {c} records in {dt * 1e3:.1f} ms = {infl / 8 / dt / 1e9:.1f} GB/s of FASTQ", flush=True)
assert tot == whole == total_rec, (tot, whole, total_rec)
print(f"{comp / 1e9:.2f} GB of BGZF = {infl / 1e9:.2f} GB of FASTQ: whole file {t_whole * 1e3:.0f} ms; 8 shards one after the other {t_all * 1e3:.0f} ms, rows add up to {tot}")
os.unlink(path)
