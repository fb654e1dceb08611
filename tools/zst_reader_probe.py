"""COUNT(*) of a .fastq.zst through the reader (EXG_TRACE=1 for the stage times): ZST_MB of FASTQ-150 in one level-3 frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from zstd_util import compress
from exon_duckdb_amd import device
from exon_duckdb_amd.reader import ShardReader
mb = int(os.environ.get("ZST_MB", "1024"))
n_rec = mb * 1_000_000 // 332
raw = device.synth_fastq(332 * n_rec)[: 332 * n_rec].cpu().numpy().tobytes()
path = "/tmp/exg_probe.fastq.zst"
open(path, "wb").write(compress(raw, 3, True))
del raw
for _ in range(3):
    t0 = time.time(); n = ShardReader(path, "fastq").count(); dt = time.time() - t0
    assert n == n_rec
    print(f"zst reader: {332 * n_rec / 1e6:.0f} MB of FASTQ, count in {dt:.3f} s = {332 * n_rec / dt / 1e9:.1f} GB/s", flush=True)
