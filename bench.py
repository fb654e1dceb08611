#!/usr/bin/env python3
"""bench.py — FASTQ records/s into DataChunk column vectors on MI355X (BASELINE.json metric).

A step = one pass of the hot path (exg_fastq_scan through the C-ABI) over one device-resident
batch of synthetic 150 bp FASTQ (332 B/record, generated in HBM by exg_synth_fastq).  N=1 runs
BASELINE configs[1] (10 GB on one MI355X).  N>1: one process per GPU (torch.distributed, RCCL),
the file is byte-range sharded — each rank scans its own 10 GB shard of an N x 10 GB file whose
cut points are NOT record aligned (1 KiB halo, global line phase from the shard-local structure,
verified by an all_gather of the newline counts) — weak scaling, no collective on the data path;
one all_reduce of the record counts per step (the COUNT(*) of config 5).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
REC = 332


def cpu_baseline(seconds_target=12.0):
    """Oracle ("port") timed on one host core, like the reference (one core per file)."""
    import numpy as np
    from oracle import pyoracle

    n_rec = 1_000_000  # 332 MB sample of the same workload
    data = pyoracle.synth_fastq(REC * n_rec)
    t0 = time.perf_counter()
    total = 0
    reps = 0
    while True:
        n, _ = pyoracle.fastq_scan_baseline(data)
        assert n == n_rec
        total += n
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or reps >= 64:
            break
    # upper bound for a CPU build that shards byte ranges over every host core (SURVEY §8 D5 ii): the same loop
    # on all cores at once, each over the whole sample
    cores = os.cpu_count() or 1
    per_thread = max(1, min(4, 512 // cores))  # ~10 s of wall clock whatever the core count
    t1 = time.perf_counter()
    total_mt = pyoracle.fastq_scan_baseline_mt(data, cores, per_thread)
    dt_mt = time.perf_counter() - t1
    return {
        "value": total / dt,
        "unit": "records/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} x {n_rec} records ({REC * n_rec / 1e6:.0f} MB synthetic FASTQ-150, in memory), "
                  f"oracle read_batch(2048)+ArrowToDuckDB loop, {dt:.1f} s",
        "all_cores": {"value": total_mt / dt_mt, "cores": cores,
                      "sample": f"{cores} threads x {per_thread} x {n_rec} records, {dt_mt:.1f} s"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gb", type=float, default=10.0, help="shard size per GPU in GB (1e9 bytes)")
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; default) | gloo (functional test of the N>1 path)")
    ap.add_argument("--single-device", action="store_true", help="test only: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from exon_duckdb_amd import abi, device, load_library

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the record scan has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"   # where the 8-byte collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    lib = load_library()

    from exon_duckdb_amd import sharding

    # file = world x args.gb GB of FASTQ-150, cut at 16-byte boundaries (NOT record boundaries)
    file_bytes = (world * int(args.gb * 1e9)) // REC * REC     # the file ends on a record boundary
    sh = sharding.plan_shards(file_bytes, world, halo=1024)[rank]
    halo, start, n_bytes = sh.halo, sh.start, sh.n_bytes
    d_in = device.synth_fastq(n_bytes, file_offset=sh.load_offset)
    cap = n_bytes // REC + 16
    scan = device.FastqScan(n_bytes, capacity_records=cap)

    # 4-line phase of the shard start, guessed from the shard's own bytes (one tiny kernel, once);
    # every step re-verifies it against the exact newline counts the scans return (all_gather).
    import ctypes as C
    ph = torch.zeros(1, dtype=torch.int32, device="cuda")
    device.check(lib.exg_fastq_guess_phase(C.c_void_p(d_in.data_ptr()), n_bytes, halo, C.c_void_p(ph.data_ptr()),
                                           device.stream_ptr()))
    guess = int(ph.item()) & 0xFFFFFFFF
    assert guess < 4, "phase guess failed on well-formed FASTQ"
    # the guess is the phase of the first line STARTING at or after the shard start; the line that
    # contains the shard's first byte is one earlier unless the shard starts exactly on a line start
    prev_is_nl = True if sh.start == 0 else bool(int(d_in[halo - 1].item()) == 10)
    first_line_index = guess if prev_is_nl else (guess - 1) % 4
    flags = (abi.EXG_F_BOF if sh.is_first else 0) | (abi.EXG_F_EOF if sh.is_last else 0)

    def step():
        scan.launch(d_in, n_bytes=n_bytes, lead=halo, first_line_index=first_line_index,
                    payload_base=0x100000000000 + sh.load_offset, flags=flags, algo=args.algo)

    total = torch.zeros(1, dtype=torch.int64, device=coll_dev)
    lines = torch.zeros(1, dtype=torch.int64, device=coll_dev)
    line_counts = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    res = scan.fetch()
    assert res.error_code == 0 and not (res.flags & abi.EXG_RF_FALLBACK), (res.error_code, res.flags)
    n_rec_local = int(res.n_records)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
        if world > 1:
            # per step: all_gather of the exact newline counts (verifies the phase guess) and the
            # COUNT(*) all_reduce — 8 bytes per rank each, no byte of the file crosses xGMI
            lines.copy_(scan.result[1:2])
            dist.all_gather(line_counts, lines)
            total.copy_(scan.result[:1])
            dist.all_reduce(total)
    barrier()
    dt = time.perf_counter() - t0
    res = scan.fetch()
    assert res.error_code == 0 and int(res.n_records) == n_rec_local
    if world > 1:
        exact = sharding.first_line_index_from_counts([int(c.item()) for c in line_counts], rank)
        assert sharding.phase_is_consistent(first_line_index % 4, exact), "phase guess contradicted by the exact counts"

    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    nrec = torch.tensor([n_rec_local], dtype=torch.int64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(nrec)
        assert int(total.item()) == int(nrec.item()), "COUNT(*) all_reduce disagrees"
    dt = float(t.item())
    total_records = int(nrec.item())
    assert total_records == file_bytes // REC, (total_records, file_bytes // REC)

    if rank == 0:
        # per-launch device time of the scan (all kernels of one exg_fastq_scan call) from HIP events
        # on the launch stream; the dominant kernel is k_fastq_fused (see profiles/)
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        avg_ms = sum(ms) / len(ms)
        achieved = n_bytes / (avg_ms * 1e-3) / 1e9
        # the same box's read-only stream rate: exg_count_newlines over the same buffer — every
        # byte leaves HBM once, nothing is written
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        ro = []
        for i in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            device.check(lib.exg_count_newlines(C.c_void_p(d_in.data_ptr()), halo, n_bytes, C.c_void_p(cnt.data_ptr()),
                                                device.stream_ptr()))
            b.record()
            torch.cuda.synchronize()
            if i:
                ro.append(a.elapsed_time(b))
        read_only = (n_bytes - halo) / (sum(ro) / len(ro) * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_fastq_fused.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch_10GB")
        out = {
            "metric": "FASTQ records/sec into DataChunks",
            "value": total_records * args.steps / dt,
            "unit": "records/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"read_fastq on {file_bytes / 1e9:.0f} GB synthetic 150 bp FASTQ "
                            f"({REC} B/record, {total_records} records), {world}x MI355X, byte-range shards",
                "file_bytes": file_bytes,
                "bytes_per_gpu": n_bytes,
                "algo": {0: "auto(fused+gated general path)", 1: "multipass", 2: "fused"}[args.algo],
                "columns": "name,description,sequence,quality_scores as duckdb::string_t + validity",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "kernel": "exg_fastq_scan launch (k_fastq_fused dominant)",
                "algorithmic_bytes_per_launch": n_bytes,
                "avg_launch_ms": avg_ms,
                "min_launch_ms": ms[0],
                "read_only_stream_GBps": read_only,
                "frac_of_read_only_stream": achieved / read_only,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
