"""Reader-level byte-range shards at size: a 4 GB FASTQ-150 file read as 8 shards (one after the other on one GPU —
on a node each rank opens its own), COUNT(*) per shard and in total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import device
from exon_duckdb_amd.reader import ShardReader
path = "/tmp/exg_shard.fastq"
n_rec = 12_000_000
nb = 332 * n_rec
with open(path, "wb") as f:
    f.write(device.synth_fastq(nb)[:nb].cpu().numpy().tobytes())
ShardReader(path, "fastq").count()
t0 = time.time(); whole = ShardReader(path, "fastq").count(); t_whole = time.time() - t0
total, t_all = 0, 0.0
for i in range(8):
    t0 = time.time(); c = ShardReader(path, "fastq", shard_index=i, shard_count=8).count(); dt = time.time() - t0
    total += c; t_all += dt
    print(f"shard {i}/8: {c} records in {dt * 1e3:.1f} ms = {nb / 8 / dt / 1e9:.1f} GB/s", flush=True)
assert total == whole == n_rec, (total, whole)
print(f"whole file: {t_whole * 1e3:.1f} ms; 8 shards one after the other: {t_all * 1e3:.1f} ms, rows add up to {total}")
os.unlink(path)
