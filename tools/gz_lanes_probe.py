"""COUNT(*) of a BGZF FASTQ-150 file (GZ_GB compressed GB, default 3) for several numbers of windows in flight and segment
sizes: what keeps the device's 8192 inflate slots full.  Run on the GPU box."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
from exon_duckdb_amd import abi, load_library
from exon_duckdb_amd.reader import ShardReader

def main():
    lib = load_library()
    gb = float(os.environ.get("GZ_GB", "3"))
    n_rec = int(gb * 1e9 * 1.93) // 332 // 16320 * 16320
    d = tempfile.mkdtemp(dir="/dev/shm")
    p = os.path.join(d, "x.fastq.gz")
    t0 = time.time()
    comp = bench.build_bgzf(abi.EXG_SYNTH_FASTQ_SEED, n_rec, p, bench.effective_cores())
    print(f"built {comp/1e9:.2f} GB of BGZF = {n_rec*332/1e9:.2f} GB of FASTQ in {time.time()-t0:.0f} s", flush=True)
    for lanes in os.environ.get("LANES", "3,2,4,6").split(","):
        for batch in os.environ.get("BATCHES", "268435456,134217728,536870912").split(","):
            os.environ["EXG_GZ_LANES"] = lanes
            best = 1e9
            for _ in range(3):
                r = ShardReader(p, "fastq", device_batch_bytes=int(batch))
                t0 = time.perf_counter()
                n = r.count()
                best = min(best, time.perf_counter() - t0)
                r.close()
                assert n == n_rec
            print(f"lanes {lanes} segment {int(batch)>>20} MiB: {best*1e3:.1f} ms = {n_rec*332/best/1e9:.1f} GB/s of FASTQ", flush=True)
    os.unlink(p); os.rmdir(d)


if __name__ == "__main__":   # (bench.build_bgzf spawns worker processes: they import this module)
    main()
