"""BASELINE config 4 alone: `build` writes C4_GB (10) GB of BGZF FASTQ-150 to /dev/shm/exg_c4.fastq.gz (bench.build_bgzf), `run`
times COUNT(*) and the all-columns drain on it (best of 3 each) under whatever environment the process was started with —
tools/ab_c4.sh loops over EXG_GZ_LANES / GPU_MAX_HW_QUEUES / library builds inside ONE box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    path = "/dev/shm/exg_c4.fastq.gz"
    gb = float(os.environ.get("C4_GB", "10"))
    if sys.argv[1:] == ["build"]:
        import bench
        from exon_duckdb_amd import abi
        n_rec = int(gb * 1e9 * 1.93) // 332
        t0 = time.perf_counter()
        comp = bench.build_bgzf(abi.EXG_SYNTH_FASTQ_SEED, n_rec, path, min(bench.effective_cores(), 192))
        print(f"built {comp / 1e9:.2f} GB of BGZF = {n_rec * 332 / 1e9:.2f} GB of FASTQ in {time.perf_counter() - t0:.0f} s", flush=True)
        sys.exit(0)
    import torch  # noqa: F401  (HIP comes up with the environment of this process)
    import bench
    from exon_duckdb_amd import load_library
    lib = load_library()
    bench.reader_count(lib, path, "fastq")
    n, dt = min((bench.reader_count(lib, path, "fastq") for _ in range(3)), key=lambda x: x[1])
    infl = n * 332
    out = f"COUNT(*) {dt * 1e3:7.1f} ms = {infl / dt / 1e9:5.1f} GB/s of FASTQ"
    if not os.environ.get("C4_COUNT_ONLY"):
        rows, chunks, dt_a = min((bench.reader_chunks(lib, path, "fastq") for _ in range(3)), key=lambda x: x[2])
        out += f" | all columns {dt_a * 1e3:7.1f} ms = {infl / dt_a / 1e9:5.1f} GB/s"
    print(out, flush=True)


if __name__ == "__main__":
    main()
