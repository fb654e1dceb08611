"""Where do the page-cache pages of the end-to-end leg's file sit, and does it matter?  Writes the 4 GB FASTQ-150 file of
bench.py's `end_to_end` leg once per NUMA node (the writing thread bound to that node's CPUs: first touch places the pages), then
times COUNT(*) and the all-columns drain through the reader on each copy.  Prints the device's node, the nodes' CPUs, the CPUs this
process may use and `numastat`-like per-node figures of the file (from /proc/self/numa_maps of a mapping).  Run on the GPU box."""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from exon_duckdb_amd import device, load_library  # noqa: E402


def cpus_of(node):
    with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
        out = []
        for part in f.read().strip().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            out.extend(range(int(a), int(b or a) + 1))
        return out


def main():
    gb = float(os.environ.get("GB", "4"))
    n = int(gb * 1e9) // 332 * 332
    allowed = sorted(os.sched_getaffinity(0))
    nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
    info = bench.rank_placement(torch, 0, 0)
    print(f"device node {info.get('numa_node')}, pci {info.get('pci')}; nodes {nodes}; this process may use {len(allowed)} CPUs: {allowed[:4]}..{allowed[-4:]}", flush=True)
    for nd in nodes:
        c = cpus_of(nd)
        print(f"  node {nd}: {len(c)} CPUs, {len(set(c) & set(allowed))} of them usable here", flush=True)
    lib = load_library()
    d = device.synth_fastq(n)[:n]
    tmp, _ = bench.scratch_dir() if hasattr(bench, "scratch_dir") else ("/dev/shm", 0)
    for nd in nodes + [None]:
        use = sorted(set(cpus_of(nd)) & set(allowed)) if nd is not None else allowed
        if not use:
            print(f"node {nd}: no usable CPU, skipped", flush=True)
            continue
        os.sched_setaffinity(0, use)
        p = os.path.join(tmp, f"numa_probe_{nd}.fastq")
        bench.write_device_bytes(torch, d, n, p)
        os.sched_setaffinity(0, allowed)
        bench.reader_count(lib, p, "fastq")
        t_count = sorted(bench.reader_count(lib, p, "fastq")[1] for _ in range(5))
        t_all = sorted(bench.reader_chunks(lib, p, "fastq")[2] for _ in range(5))
        print(f"file written from node {nd}: COUNT(*) {t_count[0]*1e3:.1f} / {t_count[2]*1e3:.1f} ms (best / median of 5), all columns {t_all[0]*1e3:.1f} / {t_all[2]*1e3:.1f} ms "
              f"= {n/t_all[0]/1e9:.1f} GB/s", flush=True)
        os.unlink(p)


if __name__ == "__main__":
    main()
