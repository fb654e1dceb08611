#!/usr/bin/env python3
"""bench.py — FASTQ records/s into DataChunk column vectors on MI355X (BASELINE.json metric).

Headline (`value`): a step = `launches_per_step` passes of the hot path (exg_fastq_scan through the C-ABI) over one
device-resident batch of synthetic 150 bp FASTQ (332 B/record, generated in HBM by exg_synth_fastq); the default 20 steps
x 20 launches keep the timed region above one second.  N=1 runs BASELINE configs[1] (10 GB on one MI355X).  N>1: one
process per GPU (torch.distributed, RCCL); the file is byte-range sharded, 12.5 GB per GPU (N=8 is configs[4]: 100 GB) —
each rank scans its own shard of the N x 12.5 GB file whose cut points are NOT record aligned (1 KiB halo, global line
phase from the shard-local structure, verified by an all_gather of the newline counts) — weak scaling, no collective on
the data path; one all_reduce of the record counts per launch (the COUNT(*) of config 5).  After the timed loop the
output of the last launch is verified on the device against the generator's closed forms (`verified`).

Beside it (rank 0, N=1, skipped with --no-configs): `configs` — BASELINE configs[0] (SELECT COUNT(*) on a 1 MB FASTA
through the reader: latency), configs[2] (8-column VCF scan, 5 GB generated in HBM by exg_synth_vcf) and configs[3]
(read_fastq on BGZF: members deflated on the host cores, inflated + scanned on the device, COUNT(*) through the reader) —
and `end_to_end` / `configs.end_to_end_vcf` (file in the page cache -> host DataChunks through exg_open / exg_next_chunk, PCIe inclusive;
never `value`).  Round 5: config 4 and the zstd leg also time what the metric names on a compressed input — records INTO DataChunks
(`all_columns`: the decoded bytes and the string_t cross PCIe back; `projected_name`: only one column's bytes do) — against the link's
rates measured in the run (`configs.pcie_link`); `end_to_end_arrow` is the reference's own boundary (new_reader -> Arrow C stream);
`roofline.traffic` comes from two rocprofv3 --pmc passes over a child of this script started before torch is imported (--no-traffic skips them).  With N>1 the same file-level leg runs sharded (`reader_sharded`: every rank opens the same file with
shard_index = rank).  `cpu_baseline` times the oracle (CPU restatement) on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import struct
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A reader keeps 6-8 HIP streams busy per device; HIP's default of 4 hardware queues makes them share (a batch's 0.1 ms scan then
# sits behind the decoder lanes' kernels and the 5 ms host copies of decoded segments: DESIGN 5.3a).  libexon_gpu sets this default
# itself when it is what brings HIP into the process; here torch initialises HIP first, so the script does (never overriding).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
REC = 332
BASE = 0x100000000000  # payload_base of the device-level launches


def cpu_baseline(seconds_target=12.0):
    """Oracle ("port") timed on one host core, like the reference (one core per file)."""
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


def measure_traffic_in_run(n_bytes_expected, timeout_s=240):
    """roofline.traffic measured IN this run: the HBM bytes one launch of the headline kernel moves, from the FETCH_SIZE and
    WRITE_SIZE counters — two `rocprofv3 --kernel-trace --pmc <counter>` passes (separate passes, no other trace domain:
    MI355X_MICROARCH.md's HBM / rocprofv3 section) over a CHILD `python3 bench.py` with a few launches of the same 10 GB
    workload, started before this process imports torch or touches the GPU.  FETCH_SIZE / WRITE_SIZE are reported in KiB; on
    gfx950 FETCH_SIZE counts half the bytes of wide coalesced streaming reads (the guide's correction): doubled.
    -> dict or None (no rocprofv3, a profiler already around this process, a pass that failed or ran out of time)."""
    import csv
    import glob
    import shutil
    import subprocess
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof or os.environ.get("EXG_BENCH_NO_TRAFFIC"):
        return None
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None   # this process is being profiled itself
    kernel = "k_fused<exg::FastqFormat, 0>"
    out = {}
    tmp = tempfile.mkdtemp(prefix="exg_pmc_", dir="/tmp")
    t0 = time.perf_counter()
    try:
        env = dict(os.environ, TMPDIR="/tmp", EXG_BENCH_NO_TRAFFIC="1")
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            cmd = [prof, "--kernel-trace", "--pmc", c, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                   "python3", os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--launches-per-step", "2", "--no-configs", "--no-cpu-baseline"]
            left = timeout_s - (time.perf_counter() - t0)
            if left < 30:
                return None
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
            if res.returncode != 0:
                return None
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if kernel in r["Kernel_Name"] and r["Counter_Name"] == c:
                            vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            out[c] = (sum(vals) / len(vals) * 1024.0, len(vals))
            # the child's line says which workload its launches ran on
            for ln in res.stdout.splitlines():
                if ln.startswith("{"):
                    out["n_bytes"] = json.loads(ln)["config"]["bytes_per_gpu"]
    except Exception:  # noqa: BLE001 (timeout, a box without counters, ...)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if out.get("n_bytes") != n_bytes_expected:
        return None
    rd, wr = out["FETCH_SIZE"][0] * 2.0, out["WRITE_SIZE"][0]
    return {"traffic": rd + wr, "read_bytes": rd, "write_bytes": wr, "dispatches": out["FETCH_SIZE"][1], "seconds": time.perf_counter() - t0,
            "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two passes over a child `python3 bench.py "
                      "--steps 1 --warmup 1 --launches-per-step 2 --no-configs` started before the timed run (KiB x 1024; FETCH_SIZE x 2: gfx950 correction); "
                      f"average of {out['FETCH_SIZE'][1]} dispatches of {kernel}"}


def verify_fastq(torch, scan, n_rec, base, first_record):
    """The columns the last launch left in HBM against the generator's closed forms, every row: record r of this buffer
    is global record k = first_record + r of the synthetic file; its four string_t hold the field lengths 15 / 10 / 150 /
    150, out-of-line pointers base + 332 k + field offset, the name prefix "SYN" + first digit of k, the inlined
    description "d:N:0:ACGT" with d = k mod 4; every description is valid."""
    k = first_record + torch.arange(n_rec, device="cuda", dtype=torch.int64)
    ok = True
    for col, (off, ln) in zip(scan.cols, [(1, 15), (17, 10), (28, 150), (181, 150)]):
        c = col[:n_rec]
        ok &= bool(((c[:, 0] & 0xFFFFFFFF) == ln).all())
        if ln > 12:
            ok &= bool((c[:, 1] == base + REC * k + off).all())
    first_digit = (k // 10 ** 11) % 10
    want_prefix = 0x53 | (0x59 << 8) | (0x4E << 16) | ((0x30 + first_digit) << 24)
    ok &= bool((((scan.cols[0][:n_rec, 0] >> 32) & 0xFFFFFFFF) == want_prefix).all())
    want_desc = (0x30 + (k & 3)) | (0x3A << 8) | (0x4E << 16) | (0x3A << 24)
    ok &= bool((((scan.cols[1][:n_rec, 0] >> 32) & 0xFFFFFFFF) == want_desc).all())
    words = (n_rec + 63) // 64
    if words > 1:
        ok &= bool((scan.validity[: words - 1] == -1).all())
    return ok


def timed_launches(torch, fn, reps, warm=2):
    """average device time of fn() in ms, HIP events on the launch stream"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0]


def link_rates(torch, n=1 << 30, reps=3):
    """what the PCIe link of this box moves, measured in this run: one pinned 1 GiB buffer <-> HBM, each direction alone (best of
    `reps`) -> (h2d GB/s, d2h GB/s).  The file-level legs are priced against these, not against a nominal figure."""
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = []
    for dst, src in ((dev, host), (host, dev)):
        best = None
        for _ in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dst.copy_(src, non_blocking=True)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None else min(best, ms)
        out.append(n / (best * 1e-3) / 1e9)
    del host, dev
    return out[0], out[1]


CHUNKS_HINT = 1  # exg_open_args.flags = EXG_OPEN_CHUNKS: chunks will be pulled — what the table function says at init_global


def open_reader(lib, path, fmt, shard=(0, 1), device_index=0, columns=0, flags=0, filters=None):
    from exon_duckdb_amd import abi
    lib.exg_open.argtypes = [C.POINTER(abi.OpenArgs), C.POINTER(C.c_void_p)]
    lib.exg_count_only.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.exg_close.argtypes = [C.c_void_p]
    a = abi.OpenArgs(path.encode(), fmt.encode(), None, 2048, device_index, 0, filters.encode() if filters else None, shard[0], shard[1], columns, flags)
    r = C.c_void_p()
    rc = lib.exg_open(C.byref(a), C.byref(r))
    assert rc == 0, lib.exg_last_error_message()
    return r


def reader_count(lib, path, fmt, shard=(0, 1), device_index=0, filters=None):
    r = open_reader(lib, path, fmt, shard, device_index, filters=filters)
    n = C.c_uint64(0)
    t0 = time.perf_counter()
    rc = lib.exg_count_only(r, C.byref(n))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.exg_last_error_message()
    lib.exg_close(r)
    return int(n.value), dt


def reader_chunks(lib, path, fmt, shard=(0, 1), device_index=0, columns=0, stats=None, filters=None):
    """every DataChunk pulled and released by a C loop of the scaffolding library (no interpreter between the chunks);
    stats: a dict that receives exg_reader_stats_of at the end of the stream (nested_ns, host_vector_bytes, ...)"""
    from exon_duckdb_amd import abi, load_test_library
    tl = load_test_library()
    r = open_reader(lib, path, fmt, shard, device_index, columns, CHUNKS_HINT, filters)
    rows, chunks = C.c_uint64(0), C.c_uint64(0)
    t0 = time.perf_counter()
    rc = tl.exon_tf_drain_chunks(r, C.byref(rows), C.byref(chunks))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.exg_last_error_message()
    if stats is not None:
        st = abi.ReaderStats()
        lib.exg_reader_stats_of.argtypes = [C.c_void_p, C.POINTER(abi.ReaderStats)]
        assert lib.exg_reader_stats_of(r, C.byref(st)) == 0
        stats.update({f: int(getattr(st, f)) for f, _ in abi.ReaderStats._fields_ if f != "reserved"})
    lib.exg_close(r)
    return int(rows.value), int(chunks.value), dt


def timed_reader_chunks(lib, path, fmt, reps=3, **kw):
    """best of `reps` drains -> (rows, chunks, seconds, the stats of that drain)"""
    best = None
    for _ in range(reps):
        st = {}
        rows, chunks, dt = reader_chunks(lib, path, fmt, stats=st, **kw)
        if best is None or dt < best[2]:
            best = (rows, chunks, dt, st)
    return best


def reader_formats_digest(lib, path, key=0):
    """read_vcf's formats column walked like an operator would (list entries, the struct's child `key` of every sample) ->
    (rows, samples, digest); the expectation is exon_tf_expect_vcf_formats_file's independent split of the file"""
    from exon_duckdb_amd import load_test_library
    tl = load_test_library()
    r = open_reader(lib, path, "vcf", columns=1 << 8, flags=CHUNKS_HINT)
    rows, samples, dg = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    rc = tl.exon_tf_drain_formats_digest(r, key, C.byref(rows), C.byref(samples), C.byref(dg))
    assert rc == 0, lib.exg_last_error_message()
    lib.exg_close(r)
    return int(rows.value), int(samples.value), int(dg.value)


def reader_digest(lib, path, fmt, want_seq_len=0, columns=0, shard=(0, 1), device_index=0, first_row=0):
    """the same walk, folding the content of EVERY row — each string_t dereferenced: length, prefix, pointer, payload bytes; VCF:
    CHROM, the parsed POS, REF — into a digest (untimed verification pass) -> (rows, chunks, digest, rows of the wrong length).
    A shard's rows are rows [first_row, first_row + n) of the file: the shards' digests add up to the file's."""
    from exon_duckdb_amd import load_test_library
    tl = load_test_library()
    r = open_reader(lib, path, fmt, shard, device_index, columns, CHUNKS_HINT)
    rows, chunks, dg, bad = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    rc = tl.exon_tf_drain_digest_from(r, {"vcf": 1, "fasta": 2}.get(fmt, 0), want_seq_len, first_row, C.byref(rows), C.byref(chunks), C.byref(dg), C.byref(bad))
    assert rc == 0, lib.exg_last_error_message()
    lib.exg_close(r)
    return int(rows.value), int(chunks.value), int(dg.value), int(bad.value)


def rank_placement(torch, rank, local_rank):
    """this rank's device and host placement: PCI bus id, the NUMA node sysfs gives for it (where the library's pinned pool
    places this device's blocks: exg_rd_io.cpp numa_node_of_device reads the same file), the CPUs the process may use"""
    info = {"rank": rank, "local_rank": local_rank, "pid": os.getpid()}
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
        info.update({"device": p.name, "pci": bdf, "cus": getattr(p, "multi_processor_count", None), "hbm_GB": round(p.total_memory / 1e9, 1)})
        try:
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                info["numa_node"] = int(f.read().strip())
        except OSError:
            info["numa_node"] = None
        info["pinned_pool_node"] = max(0, info["numa_node"]) if info["numa_node"] is not None else 0
    except Exception as e:  # noqa: BLE001
        info["error"] = f"{type(e).__name__}: {e}"
    try:
        cpus = sorted(os.sched_getaffinity(0))
        info["cpus"] = f"{cpus[0]}-{cpus[-1]} ({len(cpus)})" if cpus else ""
    except Exception:  # noqa: BLE001
        pass
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_MAX_HW_QUEUES"):
        if os.environ.get(k) is not None:
            info[k] = os.environ[k]
    return info


def effective_cores():
    """cores this process may really use: the affinity mask and the cgroup CPU quota, not the machine's core count"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def scratch_dir(need_bytes=0):
    """a directory for the file-level legs: shared memory when it has the room, else the temp dir -> (path, free bytes)"""
    import shutil
    best = None
    for d in ("/dev/shm", tempfile.gettempdir()):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            free = shutil.disk_usage(d).free
            if best is None or (best[1] < need_bytes and free > best[1]):
                best = (d, free)
    return tempfile.mkdtemp(prefix="exg_bench_", dir=best[0]), best[1]


def write_device_bytes(torch, t, n, path):
    """device tensor -> file, through 1 GiB host slices"""
    with open(path, "wb") as f:
        for o in range(0, n, 1 << 30):
            f.write(t[o:min(n, o + (1 << 30))].cpu().numpy().tobytes())


def _bgzf_worker(args):
    """deflates records [rec_lo, rec_hi) of the synthetic FASTQ-150 file into BGZF members of 65 280 bytes: the bytes come from
    the generator itself (the scaffolding library's host form of it), slice by slice — the plain file is never written"""
    seed, rec_lo, rec_hi, out_path = args
    import numpy as np
    from exon_duckdb_amd import load_test_library
    tl = load_test_library()
    step = 65280 * 83   # lcm(332, 65 280) bytes = 16 320 records = 83 members
    assert step % REC == 0 and step % 65280 == 0
    buf = np.empty(step, dtype=np.uint8)
    with open(out_path, "wb") as out:
        r = rec_lo
        while r < rec_hi:
            n = min(step // REC, rec_hi - r)
            tl.exon_tf_synth_fastq150_host(seed, r, n, buf.ctypes.data)
            raw = buf[:n * REC].tobytes()
            for o in range(0, len(raw), 65280):
                chunk = raw[o:o + 65280]
                co = zlib.compressobj(6, zlib.DEFLATED, -15)
                d = co.compress(chunk) + co.flush()
                out.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1)
                          + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
            r += n
    return os.path.getsize(out_path)


def build_bgzf(seed, n_records, out_path, workers):
    """BASELINE config 4's input: the FASTQ-150 stream cut into 65 280-byte members, each deflated (zlib level 6) with BGZF
    framing, by `workers` host processes (input preparation; not timed).  Every worker takes a run of records whose bytes are
    a multiple of 65 280, so that the members are those of one sequential bgzip run over the whole stream."""
    import multiprocessing as mp
    unit = 65280 * 83 // REC                          # 16 320 records = lcm(332, 65 280) bytes: a whole number of members
    per = max(unit, (n_records // workers + unit - 1) // unit * unit)
    jobs = [(seed, lo, min(n_records, lo + per), f"{out_path}.part{i}") for i, lo in enumerate(range(0, n_records, per))]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
        sizes = pool.map(_bgzf_worker, jobs)
    build_bgzf.pool_s = time.perf_counter() - t0
    with open(out_path, "wb") as out:
        for (_, _, _, part), sz in zip(jobs, sizes):
            with open(part, "rb") as f:  # in-kernel copy (tmpfs -> tmpfs); the part goes at once: the peak is the file + one part
                done = 0
                while done < sz:
                    done += os.sendfile(out.fileno(), f.fileno(), done, min(sz - done, 1 << 30))
            os.unlink(part)
        out.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # the BGZF end marker
    return sum(sizes) + 28


def host_pipeline_scaling(path, device_index=0, seconds=1.0):
    """The host side of the upload path with N readers at once on this one GPU's box (at 8 GPUs every reader
    copies its bytes page cache -> pinned block -> DMA, and all of them share the host's memory system): aggregate GB/s of
    N x 8 pread threads filling pinned blocks, without and with the H2D copies behind them."""
    from exon_duckdb_amd import load_test_library
    tl = load_test_library()
    out = {"what": "aggregate page cache -> pinned host memory GB/s of N readers x 8 pread threads (8 MiB slices), "
                   "`with_h2d`: each slice sent on to HBM like an upload (one GPU: the link bounds the sum); `adaptive`: with the thread "
                   "count the library picks when N uploads run at once on this many usable cores", "readers": {}}
    cores = effective_cores()
    out["usable_cores"] = cores
    for n in (1, 2, 4, 8):
        a = tl.exon_tf_host_pipeline_probe(path.encode(), n, 8, 0, device_index, seconds)
        b = tl.exon_tf_host_pipeline_probe(path.encode(), n, 8, 1, device_index, seconds)
        # what the library does by itself when n uploads run at once: min(8, max(2, usable cores / n)) threads each
        t = min(8, max(2, cores // n))
        c = tl.exon_tf_host_pipeline_probe(path.encode(), n, t, 0, device_index, seconds) if t != 8 else a
        out["readers"][str(n)] = {"pinned_GB/s": round(a / 1e9, 2) if a > 0 else None, "with_h2d_GB/s": round(b / 1e9, 2) if b > 0 else None,
                                  "threads_each_adaptive": t, "pinned_adaptive_GB/s": round(c / 1e9, 2) if c > 0 else None}
    # the same bytes WITHOUT the bounce copy (round 3's verdict asked for the measurement): hipHostRegister of 256 MiB windows of the
    # file's mapping, H2D straight from the page cache.  One reader: over one warm mapping (the windows' page-table entries exist
    # from the second pass on) and with a fresh mapping per window (what a reader that sees every byte once pays: the registration
    # makes 65 536 entries per window).  Not adopted: see DESIGN 5.
    try:
        ms1, ms3 = C.c_double(0), C.c_double(0)
        z1 = tl.exon_tf_host_zero_bounce_probe(path.encode(), 1, 1, device_index, 1.0, C.byref(ms1))
        z3 = tl.exon_tf_host_zero_bounce_probe(path.encode(), 1, 3, device_index, 1.0, C.byref(ms3))
        out["zero_bounce_1_reader"] = {"what": "hipHostRegister of 256 MiB windows of the file mapping + H2D from them (no CPU copy)",
                                       "warm_mapping_GB/s": round(z1 / 1e9, 2) if z1 > 0 else None, "warm_register_ms_per_window": round(ms1.value, 2),
                                       "fresh_mapping_per_window_GB/s": round(z3 / 1e9, 2) if z3 > 0 else None,
                                       "fresh_register_ms_per_window": round(ms3.value, 2), "adopted": False}
    except Exception as e:  # noqa: BLE001
        out["zero_bounce_1_reader"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _string_t_matches(torch, col, n_rows, off, ln, base, data_t):
    """column vector rows [0, n_rows) against (offsets, lengths) into the device buffer data_t: length word, 4-byte prefix /
    inlined bytes, pointer = base + offset — what duckdb::string_t of those slices is (exg_string_t)"""
    c = col[:n_rows]
    off_t = torch.from_numpy(off).cuda()
    ln_t = torch.from_numpy(ln).cuda()
    ok = bool(((c[:, 0] & 0xFFFFFFFF) == ln_t).all())
    # bytes 4..7: the first (up to) four bytes of the field
    pref = torch.zeros_like(ln_t)
    for j in range(4):
        pref |= torch.where(ln_t > j, data_t[off_t + j].to(torch.int64), torch.zeros_like(ln_t)) << (8 * j)
    ok &= bool((((c[:, 0] >> 32) & 0xFFFFFFFF) == pref).all())
    long_ = ln_t > 12
    ok &= bool((c[:, 1][long_] == (base + off_t)[long_]).all())
    if bool((~long_).any()):   # inlined: bytes 8..15 hold bytes 4..11 of the field, zero padded
        w = torch.zeros_like(ln_t)
        for j in range(8):
            w |= torch.where(ln_t > 4 + j, data_t[torch.clamp(off_t + 4 + j, max=data_t.numel() - 1)].to(torch.int64), torch.zeros_like(ln_t)) << (8 * j)
        ok &= bool((c[:, 1][~long_] == w[~long_]).all())
    return ok


def run_record_shapes(torch, lib, args):
    """Records of other shapes than BASELINE's 150 bp reads / 49-byte VCF lines, device resident (the
    reference's line readers run at one rate whatever the record length): HiFi-like 15 kb and ONT-like 1-100 kb reads, 36 bp
    reads, VCF lines of 100 and 2 504 samples.  A block of whole records from a seeded host generator
    (exon_duckdb_amd/testing/shapes.py, checked by the oracle in tests/) is tiled to ~args.shape_gb GB in HBM.  Timed: the
    any-shape scan alone (EXG_ALGO_FUSED_FULL) — what a reader runs from its second batch on, once the lean scan of the
    first batch came back with EXG_RF_REDO — and, beside it, that first kind of launch (EXG_ALGO_FUSED: lean scan + any-shape
    run over what it marked).  Verified: no launch gave up (EXG_RF_FALLBACK clear), the row count, and every row of the
    first, a middle and the last tile against what the generator says its rows are (lengths, prefixes, pointers)."""
    from exon_duckdb_amd import abi, device
    from exon_duckdb_amd.testing import shapes
    out = {}
    only = getattr(args, "shape_only", "")   # (tools/shapes_probe.py: legs whose name begins with this)
    target = int(args.shape_gb * 1e9)

    def tiled(header, block):
        reps = max(1, target // len(block))
        n = len(header) + reps * len(block)
        d = torch.zeros((n + 15) // 16 * 16 + 64, dtype=torch.uint8, device="cuda")
        if header:
            d[:len(header)].copy_(torch.frombuffer(bytearray(header), dtype=torch.uint8))
        d[len(header):n] = torch.frombuffer(bytearray(block), dtype=torch.uint8).cuda().repeat(reps)
        return d, n, reps

    lib.exg_scan_algo_hint.restype = C.c_int
    lib.exg_scan_algo_hint.argtypes = [C.c_int, C.c_char_p, C.c_uint64]
    algo_name = {abi.EXG_ALGO_FUSED: "EXG_ALGO_FUSED (the lean scan)", abi.EXG_ALGO_FUSED_FULL: "EXG_ALGO_FUSED_FULL (the any-shape scan)",
                 abi.EXG_ALGO_FUSED_INDEX: "EXG_ALGO_FUSED_INDEX (the any-shape scan + k_vcf_rows)"}

    def fastq_leg(name, what, block, expect):
        rows_blk = len(expect["name"][0])
        d_in, n, reps = tiled(b"", block)
        scan = device.FastqScan(n, capacity_records=rows_blk * reps + 16)
        scan.launch(d_in, payload_base=BASE, algo=abi.EXG_ALGO_FUSED)
        r1 = scan.fetch()
        # what a reader launches FIRST on this input since round 6: chosen from the first MiB on the host (exg_scan_algo_hint)
        hint = int(lib.exg_scan_algo_hint(abi.EXG_FMT_FASTQ, bytes(block[:1 << 20]), min(len(block), 1 << 20)))
        ms_lean_redo, _ = timed_launches(torch, lambda: scan.launch(d_in, payload_base=BASE, algo=abi.EXG_ALGO_FUSED), 3, warm=0)
        ms_first, _ = timed_launches(torch, lambda: scan.launch(d_in, payload_base=BASE, algo=hint), 3, warm=0)
        ms, ms_min = timed_launches(torch, lambda: scan.launch(d_in, payload_base=BASE, algo=abi.EXG_ALGO_FUSED_FULL), 6, warm=1)
        res = scan.fetch()
        ok = res.error_code == 0 and r1.error_code == 0 and int(res.n_records) == rows_blk * reps == int(r1.n_records)
        ok = ok and not ((res.flags | r1.flags) & abi.EXG_RF_FALLBACK) and bool(r1.flags & abi.EXG_RF_REDO)
        if ok:
            for m in sorted({0, reps // 2, reps - 1}):
                for k, c in enumerate(["name", "description", "sequence", "quality_scores"]):
                    col = scan.cols[k][m * rows_blk:(m + 1) * rows_blk]
                    if expect[c] is None:
                        ok = ok and not bool(col.any())
                    else:
                        off, ln = expect[c]
                        ok = ok and _string_t_matches(torch, col, rows_blk, off + m * len(block), ln, BASE, d_in)
        out[name] = {"workload": f"read_fastq, {what}: {n / 1e9:.2f} GB in HBM ({rows_blk * reps} records, a {len(block) / 1e6:.0f} MB block x {reps})",
                     "algorithmic_bytes": n, "ms": ms, "ms_min": ms_min, "GB/s": n / (ms * 1e-3) / 1e9, "frac": n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "records_per_s": rows_blk * reps / (ms * 1e-3), "algo": "EXG_ALGO_FUSED_FULL (the any-shape scan alone)",
                     "first_batch_ms": ms_first, "first_batch_GB/s": n / (ms_first * 1e-3) / 1e9,
                     "first_batch_algo": algo_name[hint] + ": exg_scan_algo_hint over the input's first MiB, what a reader launches first",
                     "lean_plus_redo_ms": ms_lean_redo,
                     "lean_plus_redo": "EXG_ALGO_FUSED on this shape (lean scan + any-shape run over the super-tiles it marked: what a reader's first batch cost before round 6)",
                     "fallback": bool((res.flags | r1.flags) & abi.EXG_RF_FALLBACK), "verified": bool(ok)}
        del scan, d_in
        torch.cuda.empty_cache()

    def vcf_leg(name, n_samples, n_lines):
        hdr, block, e = shapes.vcf_multisample_block(n_lines, n_samples, seed=n_samples)
        d_in, n, reps = tiled(hdr, block)
        scan = device.VcfScan(n, capacity_records=n_lines * reps + 16)
        kw = dict(n_bytes=n, lead=len(hdr), payload_base=BASE)
        scan.launch(d_in, algo=abi.EXG_ALGO_FUSED, **kw)
        r1 = scan.fetch()
        hint = int(lib.exg_scan_algo_hint(abi.EXG_FMT_VCF, bytes(block[:1 << 20]), min(len(block), 1 << 20)))
        ms_lean_redo, _ = timed_launches(torch, lambda: scan.launch(d_in, algo=abi.EXG_ALGO_FUSED, **kw), 3, warm=0)
        ms_first, _ = timed_launches(torch, lambda: scan.launch(d_in, algo=hint, **kw), 3, warm=0)
        ms, ms_min = timed_launches(torch, lambda: scan.launch(d_in, algo=abi.EXG_ALGO_FUSED_FULL, **kw), 6, warm=1)
        res = scan.fetch()
        # ... and with the rows left to a kernel of their own (EXG_ALGO_FUSED_INDEX: what the reader switches to on lines of >= 256 bytes);
        # the rows checked below are this launch's
        try:
            ms_ix, ms_ix_min = timed_launches(torch, lambda: scan.launch(d_in, algo=abi.EXG_ALGO_FUSED_INDEX, **kw), 6, warm=1)
            res_ix = scan.fetch()
        except Exception:  # (a library build from before the selector existed: tools/ab_shapes.sh)
            ms_ix = ms_ix_min = float("nan")
            res_ix = res
        ok = res.error_code == 0 and r1.error_code == 0 and int(res.n_records) == n_lines * reps == int(r1.n_records)
        ok = ok and res_ix.error_code == 0 and int(res_ix.n_records) == n_lines * reps
        ok = ok and not ((res.flags | r1.flags | res_ix.flags) & abi.EXG_RF_FALLBACK)
        if ok:
            pos = torch.from_numpy(e["pos"]).cuda()
            chrom = torch.from_numpy(e["chrom"]).cuda()
            want_chrom = torch.where(chrom < 10, 1 | ((0x30 + chrom) << 32), 2 | ((0x30 + chrom // 10) << 32) | ((0x30 + chrom % 10) << 40))
            for m in sorted({0, reps // 2, reps - 1}):
                sl = slice(m * n_lines, (m + 1) * n_lines)
                ok = ok and bool((scan.pos[sl] == pos).all()) and bool((scan.cols[0][sl][:, 0] == want_chrom).all())
                off, ln = e["formats"]
                ok = ok and _string_t_matches(torch, scan.cols[8][sl], n_lines, off + len(hdr) + m * len(block), ln, BASE, d_in)
            nq = int(e["qual_valid"].sum()) * reps
            words = (n_lines * reps + 63) // 64
            bits = scan.qual_valid[:words].clone()
            if (n_lines * reps) % 64:
                bits[-1] &= (1 << ((n_lines * reps) % 64)) - 1
            got = sum(int(((bits >> b) & 1).sum()) for b in range(64))
            ok = ok and got == nq
        # what the reader runs from its third batch on: lines of >= 640 B go to EXG_ALGO_FUSED_INDEX (exg_rd_batch.cpp), shorter ones stay
        use_ix = len(block) // n_lines >= 640 and ms_ix == ms_ix
        ms_full, ms_full_min = ms, ms_min
        if use_ix:
            ms, ms_min = ms_ix, ms_ix_min
        out[name] = {"workload": f"read_vcf, lines of {n_samples} samples ({len(block) // n_lines} B each): {n / 1e9:.2f} GB in HBM "
                                 f"({n_lines * reps} lines, a {len(block) / 1e6:.0f} MB block x {reps}), all columns + typed POS / QUAL",
                     "algorithmic_bytes": n, "ms": ms, "ms_min": ms_min, "GB/s": n / (ms * 1e-3) / 1e9, "frac": n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "lines_per_s": n_lines * reps / (ms * 1e-3),
                     "algo": ("EXG_ALGO_FUSED_INDEX (the any-shape scan notes where the lines end, k_vcf_rows parses the rows behind it: the reader's "
                              "choice for lines of >= 640 B)" if use_ix else "EXG_ALGO_FUSED_FULL (the any-shape scan alone)"),
                     "full_ms": ms_full, "full_GB/s": n / (ms_full * 1e-3) / 1e9,
                     "indexed_ms": ms_ix, "indexed_ms_min": ms_ix_min, "indexed_GB/s": n / (ms_ix * 1e-3) / 1e9,
                     "first_batch_ms": ms_first, "first_batch_GB/s": n / (ms_first * 1e-3) / 1e9,
                     "first_batch_algo": algo_name[hint] + ": exg_scan_algo_hint over the input's first MiB, what a reader launches first",
                     "lean_plus_redo_ms": ms_lean_redo,
                     "lean_scan_marked_tiles": bool(r1.flags & abi.EXG_RF_REDO),
                     "fallback": bool((res.flags | r1.flags) & abi.EXG_RF_FALLBACK), "verified": bool(ok)}
        del scan, d_in
        torch.cuda.empty_cache()

    def leg(fn, key, *a):
        if only and not key.startswith(only):
            return
        try:
            fn(key, *a)
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    leg(fastq_leg, "fastq_long_hifi", "HiFi-like reads of 15 kb +- 3 kb", *shapes.fastq_long_block(shapes.hifi_lengths(2000, seed=21), seed=21))
    leg(fastq_leg, "fastq_long_ont", "ONT-like reads, log-uniform 1 - 100 kb", *shapes.fastq_long_block(shapes.ont_lengths(2500, seed=22), seed=22))
    leg(fastq_leg, "fastq_short_36bp", "36 bp reads, 10-byte name lines (830 lines per 16 KiB)", *shapes.fastq_fixed_block(700000, 36, seed=23))
    leg(vcf_leg, "vcf_multisample_100", 100, 100000)
    leg(vcf_leg, "vcf_multisample_2504", 2504, 6000)
    return out


def run_configs(torch, lib, args):
    """BASELINE configs 1, 3, 4 and the end-to-end leg on one GPU -> dicts for the bench line"""
    from exon_duckdb_amd import abi, device
    out = {}
    cores = effective_cores()
    n_e2e = int(args.e2e_gb * 1e9) // REC * REC
    # config 4's input, 10 GB of BGZF = 19.3 GB of FASTQ-150, is deflated by the host's usable cores straight from the
    # generator (~22 MB/s per core at zlib level 6: about a minute on 16 cores); the plain file is never written, so the
    # scratch directory holds the 4 GB end-to-end file + the 10 GB of BGZF (+ one part while they are joined)
    budget = cores * 20e6 * args.gz_build_s
    n_gz_in = int(min(args.gz_gb * 1e9 * 1.93, budget)) // REC * REC
    tmp, free = scratch_dir(int(n_e2e + 0.6 * n_gz_in))
    if n_e2e + 0.6 * n_gz_in > 0.85 * free:
        scale = 0.85 * free / (n_e2e + 0.6 * n_gz_in)
        n_e2e, n_gz_in = int(n_e2e * scale) // REC * REC, int(n_gz_in * scale) // REC * REC
    from exon_duckdb_amd import load_test_library
    tl = load_test_library()
    try:
        link = link_rates(torch)
    except Exception:  # noqa: BLE001
        link = (float("nan"), float("nan"))
    out["pcie_link"] = {"h2d_GB/s": link[0], "d2h_GB/s": link[1], "what": "one pinned 1 GiB buffer <-> HBM, each direction alone, best of 3, measured in this run"}
    only_legs = [x for x in getattr(args, "legs", "").split(",") if x]   # (tools: --legs end_to_end_vcf,end_to_end_vcf_cohort)

    def leg(fn, *keys):
        """a side leg of the bench line: what goes wrong in it is reported in its own object(s)"""
        if only_legs and not any(k in only_legs for k in keys):
            return
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            for k in keys:
                out.setdefault(k, {"error": f"{type(e).__name__}: {e}"})
            torch.cuda.empty_cache()

    try:
        def config1():
            # ---- config 1: SELECT COUNT(*) FROM read_fasta() on a 1 MB FASTA (plumbing + latency) -----------------------
            d_fa, n_fa = device.synth_fasta(600)
            p_fa = os.path.join(tmp, "c1.fasta")
            write_device_bytes(torch, d_fa, n_fa, p_fa)
            ts = []
            for _ in range(25):
                n, dt = reader_count(lib, p_fa, "fasta")
                assert n == 600
                ts.append(dt)
            ts.sort()
            scan = device.FastaScan(n_fa)
            ms, _ = timed_launches(torch, lambda: scan.launch(d_fa, payload_base=BASE), 20)
            res = scan.fetch()
            out["config1_fasta_1MB_count"] = {
                "workload": f"SELECT COUNT(*) FROM read_fasta('{n_fa} B synthetic FASTA, 600 records'): open + upload + scan + close",
                "algorithmic_bytes": n_fa, "ms": ts[len(ts) // 2] * 1e3, "ms_min": ts[0] * 1e3, "device_scan_ms": ms,
                "GB/s": n_fa / (ts[len(ts) // 2]) / 1e9, "frac": None, "verified": bool(res.n_records == 600 and res.error_code == 0)}
            del d_fa, scan

        def config3():
            # ---- config 3: read_vcf 8-column scan, 5 GB generated in HBM (non-periodic) --------------------------------
            n_lines = int(args.vcf_gb * 1e9 / 48.65)
            d_vcf, n_vcf = device.synth_vcf(n_lines)
            hdr = bytes(d_vcf[:4096].cpu().numpy()).index(b"#CHROM")
            hdr = hdr + bytes(d_vcf[hdr:hdr + 256].cpu().numpy()).index(b"\n") + 1
            vs = device.VcfScan(n_vcf, capacity_records=n_lines + 16)
            ms, ms_min = timed_launches(torch, lambda: vs.launch(d_vcf, n_bytes=n_vcf, lead=hdr, payload_base=BASE), 8)
            res = vs.fetch()
            per_chrom = n_lines // 22 + 1
            i = torch.arange(n_lines, device="cuda", dtype=torch.int64)
            pos_ok = bool((((vs.pos[:n_lines] - 1) // 37) == (i % per_chrom)).all())   # POS = (i mod per_chrom) * 37 + 1 + (h & 31)
            chrom = i // per_chrom + 1                                               # CHROM = i / per_chrom + 1, inlined decimal
            c0 = vs.cols[0][:n_lines, 0]
            want = torch.where(chrom < 10, 1 | ((0x30 + chrom) << 32), 2 | ((0x30 + chrom // 10) << 32) | ((0x30 + chrom % 10) << 40))
            chrom_ok = bool((c0 == want).all())
            out["config3_vcf_8col"] = {
                "workload": f"read_vcf 8(+1)-column scan, {n_vcf / 1e9:.2f} GB synthetic VCF ({n_lines} lines, exg_synth_vcf), all columns + typed POS / QUAL",
                "algorithmic_bytes": n_vcf, "ms": ms, "ms_min": ms_min, "GB/s": n_vcf / (ms * 1e-3) / 1e9,
                "frac": n_vcf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "lines_per_s": n_lines / (ms * 1e-3),
                "verified": bool(res.error_code == 0 and res.n_records == n_lines and pos_ok and chrom_ok)}
            del d_vcf, vs, i, chrom, c0, want
            torch.cuda.empty_cache()

        def zstd_leg(p_fq, want):
            """SELECT COUNT(*) FROM read_fastq('x.fastq.zst'): the end-to-end leg's file as ONE zstd frame (level 3, what `zstd x.fastq`
            writes: content size and content checksum in the frame), decoded on the device in rounds beside the scan — and the
            same frame without its checksum (what the library API writes by default), where no host core has to hash the content."""
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
            import zstd_util
            with open(p_fq, "rb") as f:
                data = f.read()
            t0 = time.perf_counter()
            comp = zstd_util.compress(data, 3, True)
            t_build = time.perf_counter() - t0
            # one host core of libzstd on a piece of the same content (the reference decodes .zst through libzstd, one frame = one thread)
            piece = zstd_util.compress(data[:min(len(data), 512 * 1000 * 1000) // REC * REC], 3, False)
            z = zstd_util.lib()
            z.ZSTD_decompress.restype = C.c_size_t
            z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            n_piece = min(len(data), 512 * 1000 * 1000) // REC * REC
            dst = C.create_string_buffer(n_piece)
            t0 = time.perf_counter()
            got_n = z.ZSTD_decompress(dst, n_piece, piece, len(piece))
            t_lib = time.perf_counter() - t0
            lib_ok = got_n == n_piece and dst.raw[:4096] == data[:4096]
            # ... and one host core of XXH64 over the same bytes (what verifies a single frame's Content_Checksum: a serial chain)
            try:
                import xxhash
                t0 = time.perf_counter()
                xxhash.xxh64(dst.raw if n_piece <= (64 << 20) else memoryview(dst)[:n_piece]).digest()
                xxh_rate = n_piece / (time.perf_counter() - t0) / 1e9
            except Exception:  # noqa: BLE001
                xxh_rate = None
            del dst, piece
            n_in = len(data)
            del data
            p_zc, p_z = os.path.join(tmp, "e2e_check.fastq.zst"), os.path.join(tmp, "e2e.fastq.zst")
            with open(p_zc, "wb") as f:
                f.write(comp)
            # the same frame without Content_Checksum: bit 2 of the Frame_Header_Descriptor cleared, the last four bytes dropped
            assert comp[:4] == b"\x28\xb5\x2f\xfd" and comp[4] & 4
            with open(p_z, "wb") as f:
                f.write(comp[:4] + bytes([comp[4] & ~4]) + comp[5:-4])
            n_comp = len(comp)
            del comp
            res = {}
            for key, path in (("with_content_checksum", p_zc), ("without_checksum", p_z)):
                reader_count(lib, path, "fastq")  # warm: pools, page cache
                n, dt = min((reader_count(lib, path, "fastq") for _ in range(3)), key=lambda x: x[1])
                # what the metric names — records INTO DataChunks: all four columns pulled and released by the C drain loop; the
                # decoded bytes (the strings' payload) and 64 B of string_t per record cross PCIe back to the host
                a_rows, a_chunks, dt_a = min((reader_chunks(lib, path, "fastq") for _ in range(3)), key=lambda x: x[2])
                v_rows, v_chunks, got, bad = reader_digest(lib, path, "fastq", 150)
                d2h = n_in + 64 * a_rows + a_rows // 8
                res[key] = {"ms": dt * 1e3, "GB/s": n_in / dt / 1e9, "GB/s_compressed": (n_comp - (4 if key == "without_checksum" else 0)) / dt / 1e9,
                            "records_per_s": n / dt,
                            "all_columns": {"ms": dt_a * 1e3, "GB/s": n_in / dt_a / 1e9, "records_per_s": a_rows / dt_a, "chunks": a_chunks, "d2h_bytes": d2h,
                                            "d2h_GB/s": d2h / dt_a / 1e9, "frac_of_d2h_link": d2h / dt_a / 1e9 / link[1]},
                            "verified": bool(n == v_rows == a_rows == n_in // REC and got == want and bad == 0)}
                os.unlink(path)
            return {"workload": f"SELECT COUNT(*) FROM read_fastq('x.fastq.zst'): {n_comp / 1e9:.2f} GB = {n_in / 1e9:.2f} GB of FASTQ-150 as ONE zstd frame "
                                f"(libzstd level 3), file in the page cache; entropy stages, execution and resolve on the device in rounds of ~1 GiB, "
                                f"two rounds overlapped, beside the scan",
                    "compressed_bytes": n_comp, "algorithmic_bytes": n_comp + 2 * n_in, **res,
                    "GB/s": res["without_checksum"]["GB/s"], "ms": res["without_checksum"]["ms"], "frac": None,
                    "all_columns_GB/s": res["without_checksum"]["all_columns"]["GB/s"], "all_columns_records_per_s": res["without_checksum"]["all_columns"]["records_per_s"],
                    "link_GB/s": {"h2d": link[0], "d2h": link[1]},
                    "bound_with_checksum": "XXH64 of a single frame is one serial chain: one host core hashes it beside the decode (frames up "
                                           "to 64 MiB are hashed on the device) — that leg cannot be faster than host_xxh64_one_core_GB/s",
                    "host_xxh64_one_core_GB/s": xxh_rate,
                    "libzstd_one_core_GB/s": n_piece / t_lib / 1e9 if lib_ok else None, "input_build_s": t_build,
                    "verification": "COUNT(*) and the all-columns drain timed (best of 3 each); an untimed pass pulls all four columns as DataChunks and "
                                    "folds every row into a digest that must equal the generator's",
                    "verified": bool(all(v["verified"] for v in res.values()))}

        def arrow_leg(p_fq, want):
            """The reference's OWN boundary (boundary A: exon/include/rust.hpp:41-46): new_reader(&stream, uri, 2048, NULL, "fastq",
            NULL) -> an Arrow C stream whose record batches (Utf8 columns: int32 offsets + value bytes, built on the device and
            copied back) a C loop pulls through get_next and releases — what the reference's unchanged glue (module.cpp:228-294)
            would consume.  Timed without touching the values; an untimed pass folds every row into the generator's digest."""
            def drain(with_digest):
                rows, batches, dg = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
                err = C.create_string_buffer(512)
                t0 = time.perf_counter()
                rc = tl.exon_tf_drain_arrow_fastq(p_fq.encode(), None, None, with_digest, C.byref(rows), C.byref(batches), C.byref(dg), err, 512)
                dt = time.perf_counter() - t0
                assert rc == 0, err.value.decode("utf-8", "replace")
                return int(rows.value), int(batches.value), int(dg.value), dt
            drain(0)
            rows, batches, _, dt = min((drain(0) for _ in range(3)), key=lambda x: x[3])
            v_rows, _, got, _ = drain(1)
            # Arrow copies the VALUES back (the chunk boundary's strings point into the file's mapping): offsets + values + validity
            d2h = rows * (4 * 4 + 15 + 10 + 150 + 150) + 4 * 4 * batches
            return {"workload": f"new_reader (the reference's FFI, Arrow C stream): {n_e2e / 1e9:.1f} GB FASTQ-150 file in the page cache -> record batches of "
                                f"2048 rows on the host, all four Utf8 columns, PCIe inclusive",
                    "algorithmic_bytes": n_e2e, "ms": dt * 1e3, "GB/s": n_e2e / dt / 1e9, "records_per_s": rows / dt, "record_batches": batches,
                    "d2h_bytes": d2h, "frac": None,
                    "verification": "an untimed pass folds every row of every record batch (int32 offsets, value bytes, the description's validity "
                                    "bitmap) into a digest that must equal the generator's for these rows",
                    "verified": bool(rows == v_rows == n_e2e // REC and got == want)}

        def cold_file_leg(path, n_bytes):
            # the same file with NONE of it in the page cache (written back, then dropped with posix_fadvise(DONTNEED): what an
            # unprivileged process can do): COUNT(*) through the reader's buffered pread path — the storage's rate, not the link's
            fd = os.open(path, os.O_RDONLY)
            try:
                os.fsync(fd)
            except OSError:
                pass
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)
            n, dt = reader_count(lib, path, "fastq")
            return {"what": "COUNT(*) of the same file with its pages dropped from the page cache first (fsync + posix_fadvise DONTNEED; buffered pread, "
                            "8 threads of 8 MiB slices): bound by the scratch file system, not by PCIe", "ms": dt * 1e3, "GB/s": n_bytes / dt / 1e9,
                    "rows": n, "scratch_dir": os.path.dirname(path)}

        def files():
            # ---- end to end: FASTQ file in the page cache -> host DataChunks (PCIe inclusive) -------------------------------
            p_fq = os.path.join(tmp, "e2e.fastq")
            with open(p_fq, "wb") as f:
                step = (1 << 30) // REC * REC
                for o in range(0, n_e2e, step):
                    m = min(step, n_e2e - o)
                    f.write(device.synth_fastq(m, file_offset=o)[:m].cpu().numpy().tobytes())
            torch.cuda.empty_cache()
            reader_count(lib, p_fq, "fastq")  # warm: pools, page cache
            n, dt_c = min((reader_count(lib, p_fq, "fastq") for _ in range(3)), key=lambda x: x[1])
            rows, chunks, dt_r = min((reader_chunks(lib, p_fq, "fastq") for _ in range(3)), key=lambda x: x[2])
            # content: every row of every chunk folded into a digest, against what the generator says the rows are
            want = int(tl.exon_tf_expect_fastq150(abi.EXG_SYNTH_FASTQ_SEED, 0, n_e2e // REC, cores))
            v_rows, v_chunks, got, bad = reader_digest(lib, p_fq, "fastq", 150)
            out["end_to_end"] = {
                "workload": f"read_fastq, {n_e2e / 1e9:.1f} GB FASTQ-150 file in the page cache -> host DataChunks (exg_open / exg_next_chunk), PCIe inclusive",
                "algorithmic_bytes": n_e2e, "ms": dt_r * 1e3, "GB/s": n_e2e / dt_r / 1e9, "records_per_s": rows / dt_r, "chunks": chunks,
                "count_only_ms": dt_c * 1e3, "count_only_GB/s": n_e2e / dt_c / 1e9, "frac": None,
                "verification": "an untimed pass folds every row of every chunk (each string_t dereferenced: length, prefix, pointer, payload "
                                "bytes) into a digest that must equal the generator's for these rows",
                "frac_of_h2d_link": n_e2e / dt_r / 1e9 / link[0], "count_only_frac_of_h2d_link": n_e2e / dt_c / 1e9 / link[0],
                "link_GB/s": {"h2d": link[0], "d2h": link[1]},
                "verified": bool(rows == n == v_rows == n_e2e // REC and chunks >= (rows + 2047) // 2048 and got == want and bad == 0)}
            try:
                out["end_to_end"]["cold_file"] = cold_file_leg(p_fq, n_e2e)
            except Exception as e:  # noqa: BLE001
                out["end_to_end"]["cold_file"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["end_to_end_arrow"] = arrow_leg(p_fq, want)
            except Exception as e:  # noqa: BLE001
                out["end_to_end_arrow"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["end_to_end_zstd"] = zstd_leg(p_fq, want)
            except Exception as e:  # noqa: BLE001
                out["end_to_end_zstd"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["host_pipeline_scaling"] = host_pipeline_scaling(p_fq, 0, 0.6)
            except Exception as e:  # noqa: BLE001
                out["host_pipeline_scaling"] = {"error": f"{type(e).__name__}: {e}"}
            # ---- config 4: read_fastq on BGZF (device inflate feeding the scan) ------------------------------------------------
            p_gz = os.path.join(tmp, "c4.fastq.gz")
            t0 = time.perf_counter()
            n_gz_rec = n_gz_in // REC
            comp = build_bgzf(abi.EXG_SYNTH_FASTQ_SEED, n_gz_rec, p_gz, max(1, min(cores, 192)))
            t_build = time.perf_counter() - t0
            reader_count(lib, p_gz, "fastq")
            n, dt_g = min((reader_count(lib, p_gz, "fastq") for _ in range(3)), key=lambda x: x[1])
            # ... and what the metric names, records INTO DataChunks (module.cpp:257-294 always materialises the columns): all four
            # columns pulled and released by the C drain loop — the inflated bytes (the strings' payload) and 64 B of string_t per
            # record cross PCIe back to the host, beside the compressed bytes going up
            a_rows, a_chunks, dt_a = min((reader_chunks(lib, p_gz, "fastq") for _ in range(3)), key=lambda x: x[2])
            d2h = n_gz_in + 64 * a_rows + a_rows // 8
            # a projection: name only — the payload that crosses PCIe is the name lines' bytes, not the file's
            p_rows, p_chunks, dt_p = min((reader_chunks(lib, p_gz, "fastq", columns=0b0001) for _ in range(3)), key=lambda x: x[2])
            want = int(tl.exon_tf_expect_fastq150(abi.EXG_SYNTH_FASTQ_SEED, 0, n_gz_rec, cores))
            t1 = time.perf_counter()
            v_rows, v_chunks, got, bad = reader_digest(lib, p_gz, "fastq", 150)
            dt_v = time.perf_counter() - t1
            out["config4_fastq_bgzf"] = {
                "workload": f"SELECT COUNT(*) FROM read_fastq('x.fastq.gz'): {comp / 1e9:.2f} GB of BGZF (65 280-byte members, zlib level 6) = "
                            f"{n_gz_in / 1e9:.2f} GB of FASTQ-150, file in the page cache, inflate + scan on the device, decoded as a bounded "
                            f"stream of segments (neither the compressed nor the inflated file is resident)",
                "compressed_bytes": comp, "algorithmic_bytes": comp + 2 * n_gz_in, "ms": dt_g * 1e3, "GB/s": n_gz_in / dt_g / 1e9,
                "GB/s_compressed": comp / dt_g / 1e9, "records_per_s": n / dt_g, "frac": (comp + 2 * n_gz_in) / dt_g / 1e9 / HBM_PEAK_GBPS,
                "input_build_s": t_build, "input_deflate_pool_s": getattr(build_bgzf, "pool_s", None),
                "all_columns": {"what": "read_fastq('x.fastq.gz') into DataChunks: all four columns through exg_next_chunk / exg_release_chunk (C drain loop), best of 3",
                                "ms": dt_a * 1e3, "GB/s": n_gz_in / dt_a / 1e9, "records_per_s": a_rows / dt_a, "chunks": a_chunks,
                                "d2h_bytes": d2h, "d2h_GB/s": d2h / dt_a / 1e9, "link_GB/s": {"h2d": link[0], "d2h": link[1]},
                                "frac_of_d2h_link": d2h / dt_a / 1e9 / link[1], "upload_bound_ms": comp / link[0] / 1e6, "d2h_bound_ms": d2h / link[1] / 1e6},
                "projected_name": {"columns": "name (exg_open_args.columns = 1)", "ms": dt_p * 1e3, "GB/s": n_gz_in / dt_p / 1e9, "records_per_s": p_rows / dt_p,
                                   "chunks": p_chunks},
                "all_columns_verify_s": dt_v,
                "verification": "COUNT(*) and the all-columns drain timed (best of 3 each); an untimed pass pulls all four columns as DataChunks and folds "
                                "every row into a digest that must equal the generator's",
                "verified": bool(n == v_rows == a_rows == p_rows == n_gz_rec and got == want and bad == 0)}

        def vcf_file():
            # ---- read_vcf end to end: VCF-8 file in the page cache -> host DataChunks, every column of the reference's schema
            # (nested alt / filter / info included).  DuckDB's vectors are ~3.3x the text, so this leg is bound by the D2H link
            for f in os.listdir(tmp):                      # the FASTQ legs' files are done with
                os.unlink(os.path.join(tmp, f))
            n_lines = int(min(args.vcf_gb, 2.0) * 1e9 / 48.65)
            d_vcf, n_vcf = device.synth_vcf(n_lines)
            p_vcf = os.path.join(tmp, "e2e.vcf")
            write_device_bytes(torch, d_vcf, n_vcf, p_vcf)
            del d_vcf
            torch.cuda.empty_cache()
            reader_count(lib, p_vcf, "vcf")
            n, dt_c = min((reader_count(lib, p_vcf, "vcf") for _ in range(3)), key=lambda x: x[1])
            rows, chunks, dt_r, st_r = timed_reader_chunks(lib, p_vcf, "vcf")
            # content: CHROM, the parsed POS and REF of every row, against an independent split of the file's lines
            e_rows, e_dg = C.c_uint64(0), C.c_uint64(0)
            assert tl.exon_tf_expect_vcf_file(p_vcf.encode(), C.byref(e_rows), C.byref(e_dg)) == 0
            v_rows, v_chunks, got, _ = reader_digest(lib, p_vcf, "vcf")
            # the same scan with a projection pushed into the reader (exg_open_args.columns; DuckDB's projection_pushdown):
            # chrom, pos, ref — every column is still parsed and typed on the device, three of nine cross PCIe
            proj = 0b1011
            p_rows, p_chunks, dt_p, st_p = timed_reader_chunks(lib, p_vcf, "vcf", columns=proj)
            pv_rows, _, p_got, _ = reader_digest(lib, p_vcf, "vcf", columns=proj)
            nested_s = st_r["nested_ns"] * 1e-9
            d2h = st_r["host_vector_bytes"]
            # the reference's OWN boundary on the same file: new_reader -> Arrow C stream with the nested columns as Arrow arrays
            # (List<Utf8>, Struct, List<Struct>: converted from the DuckDB layouts on the device), every record batch pulled and released in C
            def arrow_drain(with_digest):
                rows_a, nb, dg, el = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
                err = C.create_string_buffer(512)
                t0 = time.perf_counter()
                rc = tl.exon_tf_drain_arrow_vcf(p_vcf.encode(), None, None, with_digest, C.byref(rows_a), C.byref(nb), C.byref(dg), C.byref(el), err, 512)
                dt = time.perf_counter() - t0
                assert rc == 0, err.value
                return int(rows_a.value), int(nb.value), int(dg.value), int(el.value), dt
            try:
                arrow_drain(0)
                a_rows, a_batches, _, a_elems, dt_a = min((arrow_drain(0) for _ in range(3)), key=lambda x: x[4])
                av_rows, _, a_dg, _, _ = arrow_drain(1)
                arrow_vcf = {"what": "new_reader(file_format='vcf') -> Arrow C stream, the reference's schema with its nested arrays, every record batch pulled "
                                     "and released by a C loop (PCIe inclusive; the nested columns are converted to Arrow's layouts on the device)",
                             "ms": dt_a * 1e3, "GB/s": n_vcf / dt_a / 1e9, "records_per_s": a_rows / dt_a, "record_batches": a_batches, "list_elements": a_elems,
                             "verified": bool(a_rows == av_rows == n_lines and a_dg == int(e_dg.value))}
            except Exception as e:  # noqa: BLE001
                arrow_vcf = {"error": f"{type(e).__name__}: {e}"}
            # the same file bgzip'd (x.vcf.gz, BGZF members like `bgzip` writes: what VCFs are shipped as): inflated on the device, the
            # nested columns made from the inflated text, which goes back to the host beside the vectors (the strings point into it)
            try:
                from exon_duckdb_amd.testing.bgzf import bgzip
                p_gz = p_vcf + ".gz"
                nz = bgzip(p_vcf, p_gz)
                reader_count(lib, p_gz, "vcf")
                g_n, g_dt_c = min((reader_count(lib, p_gz, "vcf") for _ in range(3)), key=lambda x: x[1])
                g_rows, g_chunks, g_dt, g_st = timed_reader_chunks(lib, p_gz, "vcf")
                gv_rows, _, g_got, _ = reader_digest(lib, p_gz, "vcf")
                os.unlink(p_gz)
                g_d2h = g_st["host_vector_bytes"] + n_vcf
                bgzip_vcf = {"what": "the same file as BGZF (zlib level 1, 65 280-byte members): inflate + scan + nested columns on the device -> host DataChunks of "
                                     "all columns; d2h_bytes = the vectors + the inflated text the strings point into",
                             "compressed_bytes": nz, "ms": g_dt * 1e3, "GB/s": n_vcf / g_dt / 1e9, "records_per_s": g_rows / g_dt, "chunks": g_chunks,
                             "count_only_ms": g_dt_c * 1e3, "count_only_GB/s": n_vcf / g_dt_c / 1e9,
                             "d2h_bytes": g_d2h, "frac_of_d2h_link": g_d2h / g_dt / 1e9 / link[1],
                             "verified": bool(g_rows == g_n == gv_rows == n_lines and g_got == int(e_dg.value))}
            except Exception as e:  # noqa: BLE001
                bgzip_vcf = {"error": f"{type(e).__name__}: {e}"}
            out["end_to_end_vcf"] = {
                "arrow_boundary": arrow_vcf,
                "bgzip": bgzip_vcf,
                # the nested columns' chain on the device (exg_vcf_nested.hip: counting passes, prefix sums, children — wall time on
                # the reader's thread up to the point where the vectors start for the host, launches and the two syncs included)
                "nested": {"what": "id / alt / filter LIST(VARCHAR), info STRUCT, formats LIST(STRUCT) of every batch made on the device: counts, "
                                   "prefix sums, children (reader-thread wall time, exg_reader_stats.nested_ns)",
                           "ms": nested_s * 1e3, "device_ms_per_GB": nested_s * 1e3 / (n_vcf / 1e9), "GB/s": n_vcf / nested_s / 1e9 if nested_s else None,
                           "frac": n_vcf / nested_s / 1e9 / HBM_PEAK_GBPS if nested_s else None,
                           "ms_when_projected_away": st_p["nested_ns"] * 1e-6},
                "d2h_bytes": d2h, "d2h_GB/s": d2h / dt_r / 1e9, "frac_of_d2h_link": d2h / dt_r / 1e9 / link[1], "d2h_bound_ms": d2h / link[1] / 1e6,
                "link_GB/s": {"h2d": link[0], "d2h": link[1]},
                "projected": {"columns": "chrom, pos, ref (exg_open_args.columns)", "ms": dt_p * 1e3, "GB/s": n_vcf / dt_p / 1e9,
                              "records_per_s": p_rows / dt_p, "verified": bool(p_rows == pv_rows == n_lines and p_got == int(e_dg.value))},
                "workload": f"read_vcf, {n_vcf / 1e9:.2f} GB VCF-8 file in the page cache -> host DataChunks of all 8 columns (exg_open / exg_next_chunk), PCIe inclusive",
                "algorithmic_bytes": n_vcf, "ms": dt_r * 1e3, "GB/s": n_vcf / dt_r / 1e9, "records_per_s": rows / dt_r, "chunks": chunks,
                "count_only_ms": dt_c * 1e3, "count_only_GB/s": n_vcf / dt_c / 1e9, "frac": None,
                "verification": "an untimed pass folds CHROM, the parsed POS and REF of every row into a digest that must equal the one of "
                                "an independent line / tab split of the file",
                "verified": bool(rows == n == n_lines == v_rows == int(e_rows.value) and chunks >= (rows + 2047) // 2048 and got == int(e_dg.value))}

        def cohort_files():
            # ---- read_vcf on COHORT lines with the reference's real schema (module.cpp:126-147: formats = LIST(STRUCT(<##FORMAT keys>))):
            # 100 and 2 504 samples a line, header with typed INFO keys and FORMAT GT, file in the page cache -> host DataChunks.
            # Every sample is a list element with a 16-byte string_t child: the vectors are 4x the text, the leg is the D2H link's
            from exon_duckdb_amd.testing import shapes
            res = {}
            for n_samples, n_lines in ((100, 100000), (2504, 6000)):
                _, block, _ = shapes.vcf_multisample_block(n_lines, n_samples, seed=n_samples)
                hdr = shapes.vcf_cohort_header(n_samples)
                reps = max(1, int(min(args.vcf_gb, 1.0) * 1e9) // len(block))
                p_c = os.path.join(tmp, f"cohort{n_samples}.vcf")
                with open(p_c, "wb") as f:
                    f.write(hdr)
                    for _ in range(reps):
                        f.write(block)
                n_c = len(hdr) + reps * len(block)
                reader_count(lib, p_c, "vcf")
                n, dt_c = min((reader_count(lib, p_c, "vcf") for _ in range(3)), key=lambda x: x[1])
                rows, chunks, dt_a, st_a = timed_reader_chunks(lib, p_c, "vcf")
                f_rows, _, dt_f, st_f = timed_reader_chunks(lib, p_c, "vcf", columns=1 << 8)
                p_rows, _, dt_p, st_p = timed_reader_chunks(lib, p_c, "vcf", columns=0b1011)
                e_rows, e_smp, e_dg = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
                assert tl.exon_tf_expect_vcf_formats_file(p_c.encode(), b"GT", C.byref(e_rows), C.byref(e_smp), C.byref(e_dg)) == 0
                v_rows, v_smp, v_dg = reader_formats_digest(lib, p_c, key=0)
                os.unlink(p_c)
                nested_s = st_a["nested_ns"] * 1e-9

                def leg_of(dt, st):
                    d2h = st["host_vector_bytes"]
                    return {"ms": dt * 1e3, "GB/s": n_c / dt / 1e9, "lines_per_s": n_lines * reps / dt, "samples_per_s": n_lines * reps * n_samples / dt,
                            "d2h_bytes": d2h, "d2h_GB/s": d2h / dt / 1e9, "frac_of_d2h_link": d2h / dt / 1e9 / link[1], "d2h_bound_ms": d2h / link[1] / 1e6}
                res[f"vcf_multisample_{n_samples}"] = {
                    "workload": f"read_vcf, {n_c / 1e9:.2f} GB file of lines with {n_samples} samples ({len(block) // n_lines} B each; {n_lines * reps} lines, "
                                f"{n_lines * reps * n_samples} samples; header: 6 typed INFO keys, FORMAT GT) in the page cache -> host DataChunks, PCIe inclusive",
                    "algorithmic_bytes": n_c, "count_only_ms": dt_c * 1e3, "count_only_GB/s": n_c / dt_c / 1e9,
                    "all_columns": leg_of(dt_a, st_a), "formats_only": leg_of(dt_f, st_f), "chrom_pos_ref": leg_of(dt_p, st_p),
                    "nested": {"ms": nested_s * 1e3, "device_ms_per_GB": nested_s * 1e3 / (n_c / 1e9), "GB/s": n_c / nested_s / 1e9 if nested_s else None,
                               "frac": n_c / nested_s / 1e9 / HBM_PEAK_GBPS if nested_s else None},
                    "link_GB/s": {"h2d": link[0], "d2h": link[1]}, "chunks": chunks, "frac": None,
                    "verification": "an untimed pass walks formats like an operator would (list entries, the struct's GT child of every sample) and folds every "
                                    "sample's GT into a digest that must equal the one of an independent line / tab / colon split of the file",
                    "verified": bool(rows == n == f_rows == p_rows == v_rows == int(e_rows.value) == n_lines * reps and v_smp == int(e_smp.value) == rows * n_samples
                                     and v_dg == int(e_dg.value))}
            out["end_to_end_vcf_cohort"] = res

        def long_reads_file():
            # ---- read_fastq end to end on LONG reads (HiFi-like 15 kb): the first device batch comes back marked by the lean scan
            # (EXG_RF_REDO), the reader switches to the any-shape scan for the rest of the file (exg_reader_stats.scan_algo)
            from exon_duckdb_amd.testing import shapes
            block, _ = shapes.fastq_long_block(shapes.hifi_lengths(2000, seed=31), seed=31)
            reps = max(1, int(min(args.e2e_gb, 2.0) * 1e9) // len(block))
            p_lr = os.path.join(tmp, "long.fastq")
            with open(p_lr, "wb") as f:
                for _ in range(reps):
                    f.write(block)
            n_lr = reps * len(block)
            reader_count(lib, p_lr, "fastq")
            n, dt_c = min((reader_count(lib, p_lr, "fastq") for _ in range(3)), key=lambda x: x[1])
            rows, chunks, dt_r = min((reader_chunks(lib, p_lr, "fastq") for _ in range(3)), key=lambda x: x[2])
            e_rows, e_dg = C.c_uint64(0), C.c_uint64(0)
            assert tl.exon_tf_expect_fastq_file(p_lr.encode(), C.byref(e_rows), C.byref(e_dg)) == 0
            v_rows, _, got, _ = reader_digest(lib, p_lr, "fastq")
            os.unlink(p_lr)
            out["end_to_end_long_reads"] = {
                "workload": f"read_fastq, {n_lr / 1e9:.1f} GB file of HiFi-like reads (15 kb +- 3 kb, {2000 * reps} records) in the page cache -> host "
                            f"DataChunks, PCIe inclusive; the reader's sticky choice: lean scan on the first batch, any-shape scan behind it",
                "algorithmic_bytes": n_lr, "ms": dt_r * 1e3, "GB/s": n_lr / dt_r / 1e9, "records_per_s": rows / dt_r, "chunks": chunks,
                "count_only_ms": dt_c * 1e3, "count_only_GB/s": n_lr / dt_c / 1e9, "frac": None,
                "verification": "an untimed pass folds every row of every chunk into a digest that must equal the one of an independent four-line "
                                "split of the file",
                "verified": bool(rows == n == v_rows == int(e_rows.value) == 2000 * reps and got == int(e_dg.value))}

        def fasta_file():
            # ---- read_fasta end to end at size: 60-column wrapped records of 0.3 - 3 kB, every third one without a description;
            # the sequences come back as the newline-free strings the device compacted (k_fa_fused), not as slices of the file
            n_rec = max(1000, int(min(args.e2e_gb, 2.0) * 1e9 / 1712))
            d_fa, n_fa = device.synth_fasta(n_rec)
            p_fa = os.path.join(tmp, "e2e.fasta")
            write_device_bytes(torch, d_fa, n_fa, p_fa)
            del d_fa
            torch.cuda.empty_cache()
            reader_count(lib, p_fa, "fasta")
            n, dt_c = min((reader_count(lib, p_fa, "fasta") for _ in range(3)), key=lambda x: x[1])
            rows, chunks, dt_r, st = timed_reader_chunks(lib, p_fa, "fasta")
            e_rows, e_dg = C.c_uint64(0), C.c_uint64(0)
            tl.exon_tf_expect_fasta_file.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            assert tl.exon_tf_expect_fasta_file(p_fa.encode(), C.byref(e_rows), C.byref(e_dg)) == 0
            v_rows, _, got, _ = reader_digest(lib, p_fa, "fasta")
            try:   # the same file bgzip'd (x.fasta.gz)
                from exon_duckdb_amd.testing.bgzf import bgzip
                p_gz = p_fa + ".gz"
                nz = bgzip(p_fa, p_gz)
                reader_count(lib, p_gz, "fasta")
                g_n, g_dt_c = min((reader_count(lib, p_gz, "fasta") for _ in range(3)), key=lambda x: x[1])
                g_rows, g_chunks, g_dt, g_st = timed_reader_chunks(lib, p_gz, "fasta")
                gv_rows, _, g_got, _ = reader_digest(lib, p_gz, "fasta")
                os.unlink(p_gz)
                bgzip_fa = {"what": "the same file as BGZF (zlib level 1, 65 280-byte members): inflate + scan on the device -> host DataChunks; the joined "
                                    "sequences and a side buffer of the definition lines' strings travel back, not the inflated text",
                            "compressed_bytes": nz, "ms": g_dt * 1e3, "GB/s": n_fa / g_dt / 1e9, "records_per_s": g_rows / g_dt, "chunks": g_chunks,
                            "count_only_ms": g_dt_c * 1e3, "count_only_GB/s": n_fa / g_dt_c / 1e9, "d2h_bytes": int(g_st.get("host_vector_bytes", 0)) or None,
                            "frac_of_d2h_link": (g_st.get("host_vector_bytes", 0) / g_dt / 1e9 / link[1]) or None,
                            "verified": bool(g_rows == g_n == gv_rows == n_rec and g_got == int(e_dg.value))}
            except Exception as e:  # noqa: BLE001
                bgzip_fa = {"error": f"{type(e).__name__}: {e}"}
            os.unlink(p_fa)
            d2h = int(st.get("host_vector_bytes", 0))
            out["end_to_end_fasta"] = {
                "workload": f"read_fasta, {n_fa / 1e9:.2f} GB file of {n_rec} records (sequences of 241 - 3 000 bases in lines of 60) in the page cache -> host "
                            f"DataChunks of id, description, sequence (the sequences joined on the device), PCIe inclusive both ways",
                "algorithmic_bytes": n_fa, "ms": dt_r * 1e3, "GB/s": n_fa / dt_r / 1e9, "records_per_s": rows / dt_r, "chunks": chunks,
                "count_only_ms": dt_c * 1e3, "count_only_GB/s": n_fa / dt_c / 1e9, "frac": None,
                "d2h_bytes": d2h or None, "frac_of_h2d_link": n_fa / dt_r / 1e9 / link[0], "frac_of_d2h_link": (d2h / dt_r / 1e9 / link[1]) if d2h else None,
                "link_GB/s": {"h2d": link[0], "d2h": link[1]}, "bgzip": bgzip_fa,
                "verification": "an untimed pass folds every row of every chunk (id, description or NULL, the joined sequence) into a digest that must "
                                "equal the one of an independent split of the file at its '>' lines",
                "verified": bool(rows == n == v_rows == int(e_rows.value) == n_rec and got == int(e_dg.value))}

        leg(config1, "config1_fasta_1MB_count")
        leg(fasta_file, "end_to_end_fasta")
        leg(config3, "config3_vcf_8col")
        leg(files, "end_to_end", "config4_fastq_bgzf")
        if "host_pipeline_scaling" in out and "end_to_end" in out and "error" not in out["end_to_end"]:
            out["end_to_end"]["host_pipeline_scaling"] = out.pop("host_pipeline_scaling")
        leg(long_reads_file, "end_to_end_long_reads")
        leg(vcf_file, "end_to_end_vcf")
        leg(cohort_files, "end_to_end_vcf_cohort")
    finally:
        for f in os.listdir(tmp):
            os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--launches-per-step", type=int, default=24, help="scan launches per step (keeps the timed region above 1 s: 20 steps x 24 x 2.3 ms)")
    ap.add_argument("--gb", type=float, default=None, help="shard size per GPU in GB (1e9 bytes); default 10 (N=1), 12.5 (N>1: N=8 is 100 GB)")
    ap.add_argument("--algo", type=int, default=2, help="2 = the fused scan, which is what the reader launches per batch (EXG_RF_FALLBACK asserted clear); 3 = the any-shape "
                    "scan alone (what a reader switches to on long / very short reads); 0 = fused + the gated general-path launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs / end_to_end legs")
    ap.add_argument("--legs", default="", help="only these configs legs (comma separated keys of the line's `configs` object; default: all)")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes over a child, ~40 s)")
    ap.add_argument("--vcf-gb", type=float, default=5.0)
    ap.add_argument("--e2e-gb", type=float, default=4.0)
    ap.add_argument("--gz-gb", type=float, default=10.0, help="config 4: compressed GB asked for (BASELINE: 10; bounded by the scratch space and --gz-build-s)")
    ap.add_argument("--gz-build-s", type=float, default=75.0, help="config 4: seconds of host deflate the input may cost (usable cores x ~20 MB/s each)")
    ap.add_argument("--shape-gb", type=float, default=4.0, help="record_shapes legs: GB of each shape tiled in HBM")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; default) | gloo (functional test of the N>1 path)")
    ap.add_argument("--single-device", action="store_true", help="test only: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # --gpus N is the number of ranks.  Without a launcher (no WORLD_SIZE in the environment) and N > 1 this process starts the N
    # ranks itself — as a CHILD `python -m torch.distributed.run` with the same arguments, before torch is imported or any
    # GPU call is made here (never an exec of a process that touched the GPU) — passes the ranks' output through and exits with
    # the child's code.  Under a launcher the world must be what --gpus says.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: start it without a launcher (it launches {args.gpus} ranks "
              f"itself) or with --nproc-per-node {args.gpus}", file=sys.stderr)
        raise SystemExit(2)

    # roofline.traffic, measured by two counter passes over a child of this script BEFORE this process touches the GPU
    # (N = 1 at the headline size only; ~40 s; --no-traffic skips it and the line falls back to the committed counters)
    traffic_run = None
    if world_env == 1 and not args.no_traffic and args.gb is None and args.algo == 2:
        traffic_run = measure_traffic_in_run(int(10.0 * 1e9) // REC * REC)

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
    # what the collective layer itself says it is (not what the command line asked for): the backend's name and the number
    # of ranks in the process group that ran the all_gather / all_reduce of this line
    coll = ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl": args.backend == "nccl",
             "distinct_devices": not args.single_device} if world > 1 else {"backend": None, "world_size": 1})
    if world > 1:
        assert dist.get_world_size() == args.gpus
        # where every rank runs — so that the first run on eight devices can be read from its JSON: the device's PCI address and
        # the NUMA node it hangs off (sysfs), the CPUs this process may run on, its per-launch time (filled in below)
        coll["ranks"] = [None] * world
        dist.all_gather_object(coll["ranks"], rank_placement(torch, rank, local_rank))

    from exon_duckdb_amd import sharding

    gb = args.gb if args.gb is not None else (10.0 if world == 1 else 12.5)
    L = max(1, args.launches_per_step)
    # file = world x gb GB of FASTQ-150, cut at 16-byte boundaries (NOT record boundaries)
    file_bytes = int(world * gb * 1e9) // REC * REC     # the file ends on a record boundary
    sh = sharding.plan_shards(file_bytes, world, halo=1024)[rank]
    halo, n_bytes = sh.halo, sh.n_bytes
    d_in = device.synth_fastq(n_bytes, file_offset=sh.load_offset)
    cap = n_bytes // REC + 16
    scan = device.FastqScan(n_bytes, capacity_records=cap)

    # 4-line phase of the shard start, guessed from the shard's own bytes (one tiny kernel, once);
    # every launch re-verifies it against the exact newline counts the scans return (all_gather).
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
    payload_base = BASE + sh.load_offset

    def launch():
        scan.launch(d_in, n_bytes=n_bytes, lead=halo, first_line_index=first_line_index,
                    payload_base=payload_base, flags=flags, algo=args.algo)

    total = torch.zeros(1, dtype=torch.int64, device=coll_dev)
    lines = torch.zeros(1, dtype=torch.int64, device=coll_dev)
    line_counts = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]

    def step():
        for _ in range(L):
            launch()
            if world > 1:
                # per launch: all_gather of the exact newline counts (verifies the phase guess) and the
                # COUNT(*) all_reduce — 8 bytes per rank each, no byte of the file crosses xGMI
                lines.copy_(scan.result[1:2])
                dist.all_gather(line_counts, lines)
                total.copy_(scan.result[:1])
                dist.all_reduce(total)

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

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    res = scan.fetch()
    assert res.error_code == 0 and int(res.n_records) == n_rec_local and not (res.flags & abi.EXG_RF_FALLBACK)
    if world > 1:
        exact = sharding.first_line_index_from_counts([int(c.item()) for c in line_counts], rank)
        assert sharding.phase_is_consistent(first_line_index % 4, exact), "phase guess contradicted by the exact counts"
    # per-launch device time of the scan from HIP events on the launch stream, outside the wall-clock region (recording
    # 2 x steps x launches events inside it would sit in the measured loop); the dominant kernel is k_fused<FastqFormat>
    avg_ms, min_ms = timed_launches(torch, launch, 20, warm=0)

    # the columns the last launch left in HBM, every row, against the generator's closed forms: records owned by this
    # shard are those whose LAST line ends in it
    verified = verify_fastq(torch, scan, n_rec_local, BASE, _first_owned_record(sh))

    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    nrec = torch.tensor([n_rec_local], dtype=torch.int64, device=coll_dev)
    ver = torch.tensor([1 if verified else 0], dtype=torch.int64, device=coll_dev)
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "timed_s": dt, "avg_launch_ms": avg_ms, "records": n_rec_local})
        for pr in per_rank:
            if coll.get("ranks") and coll["ranks"][pr["rank"]] is not None:
                coll["ranks"][pr["rank"]].update({k: v for k, v in pr.items() if k != "rank"})
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(nrec)
        dist.all_reduce(ver, op=dist.ReduceOp.MIN)
        assert int(total.item()) == int(nrec.item()), "COUNT(*) all_reduce disagrees"
    dt = float(t.item())
    total_records = int(nrec.item())
    assert total_records == file_bytes // REC, (total_records, file_bytes // REC)
    verified = bool(int(ver.item()))
    assert verified, "the scan's output differs from the generator's closed forms"

    # reader level, sharded: every rank opens the same file with its rank (file -> COUNT(*) per shard, PCIe inclusive)
    reader_sharded = None
    if world > 1 and not args.no_configs:
        reader_sharded = sharded_reader_leg(torch, dist, lib, device, world, rank, local_rank, coll_dev, args)

    if rank == 0:
        achieved = n_bytes / (avg_ms * 1e-3) / 1e9
        # the same box's read-only stream rate: exg_count_newlines over the same buffer — every
        # byte leaves HBM once, nothing is written
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        ro_ms, _ = timed_launches(torch, lambda: device.check(lib.exg_count_newlines(C.c_void_p(d_in.data_ptr()), halo, n_bytes,
                                                                                     C.c_void_p(cnt.data_ptr()), device.stream_ptr())), 5, warm=1)
        read_only = (n_bytes - halo) / (ro_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_fastq_fused.json")
        if traffic_run:
            traffic, traffic_src = traffic_run["traffic"], traffic_run["source"]
        elif os.path.exists(pmc):
            with open(pmc) as f:
                j = json.load(f)
            # (counters were taken on the 10 GB launch of config 2: only that launch is priced with them)
            traffic = j.get("hbm_bytes_per_launch_10GB") if abs(n_bytes - 9999999692) < 1e8 else None
            age_h = (time.time() - os.path.getmtime(pmc)) / 3600.0
            traffic_src = (f"profiles/pmc_fastq_fused.json, tag {j.get('tag')}, file {age_h:.1f} h old (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                           "command, gfx950 corrections; NOT re-measured in this run: --no-traffic, no rocprofv3, or the in-run passes failed)")
        out = {
            "metric": "FASTQ records/sec into DataChunks",
            "value": total_records * L * args.steps / dt,
            "unit": "records/s",
            "n_gpus": world,
            "collectives": coll,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "verified": verified,
            "config": {
                "workload": f"read_fastq on {file_bytes / 1e9:.0f} GB synthetic 150 bp FASTQ "
                            f"({REC} B/record, {total_records} records), {world}x MI355X, byte-range shards",
                "file_bytes": file_bytes,
                "bytes_per_gpu": n_bytes,
                "launches_per_step": L,
                "ms_per_launch": dt / args.steps / L * 1e3,
                "algo": {0: "auto(fused+gated general path)", 1: "multipass", 2: "fused (the reader's launch: the lean scan + the any-shape run over what it "
                         "marked — nothing on this input; EXG_RF_FALLBACK asserted clear)", 3: "fused_full (the any-shape scan alone)"}[args.algo],
                "columns": "name,description,sequence,quality_scores as duckdb::string_t + validity",
                "verification": "every row of the last launch's four columns + validity against the generator's closed forms, on the device",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_read_bytes": traffic_run["read_bytes"] if traffic_run else None,
                "traffic_write_bytes": traffic_run["write_bytes"] if traffic_run else None,
                "traffic_passes_s": traffic_run["seconds"] if traffic_run else None,
                "kernel": "exg_fastq_scan launch (k_fused<FastqFormat> dominant)",
                "algorithmic_bytes_per_launch": n_bytes,
                "avg_launch_ms": avg_ms,
                "min_launch_ms": min_ms,
                "read_only_stream_GBps": read_only,
                "frac_of_read_only_stream": achieved / read_only,
            },
        }
        if reader_sharded:
            out["reader_sharded"] = reader_sharded
        if world == 1 and not args.no_configs:
            del scan, d_in
            torch.cuda.empty_cache()
            # side legs: a failure in one of them (no scratch space, ...) is reported in its object, never at the cost of the line
            try:
                cfg = run_configs(torch, lib, args)
            except Exception as e:  # noqa: BLE001
                cfg = {"end_to_end": {"error": f"{type(e).__name__}: {e}"}, "error": f"{type(e).__name__}: {e}"}
            out["end_to_end"] = cfg.pop("end_to_end", None)
            out["end_to_end_arrow"] = cfg.pop("end_to_end_arrow", None)
            out["configs"] = cfg
            try:
                if not args.legs or "record_shapes" in args.legs.split(","):
                    out["record_shapes"] = run_record_shapes(torch, lib, args)
            except Exception as e:  # noqa: BLE001
                out["record_shapes"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n, argv):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <argv>` as a child process (this process has not
    imported torch and never touches the GPU) -> its exit code; the ranks' stdout / stderr are this process's own"""
    import socket
    import subprocess
    with socket.socket() as s:          # a free port on the loopback (the container's hostname may not resolve)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def _first_owned_record(sh):
    """global index of the first record whose last line ends at or after the shard's first byte (records are 332 bytes:
    record k ends at byte 332 (k + 1) - 1)"""
    if sh.is_first:
        return 0
    return sh.start // REC  # the record that holds byte `start` ends at or after it


def sharded_reader_leg(torch, dist, lib, device, world, rank, local_rank, coll_dev, args):
    """N ranks, one file: rank 0 writes world x e2e_gb GB of FASTQ-150 to shared memory (less when it has not the room),
    every rank opens it with shard_index = rank on its own device and counts its shard; value = rows of all shards /
    slowest rank's time.  A side leg: whatever goes wrong in it is reported in its own object and never takes the headline
    down (every rank reaches every collective)."""
    tmp = None
    plan = [None, 0, ""]          # path, bytes, why not
    if rank == 0:
        try:
            want = int(world * args.e2e_gb * 1e9) // REC * REC
            tmp, free = scratch_dir(want)
            n_file = min(want, int(0.45 * free)) // REC * REC
            if n_file < world * (64 << 20):
                plan[2] = f"no room for the shared file ({free / 1e9:.1f} GB free)"
            else:
                plan[0], plan[1] = os.path.join(tmp, "sharded.fastq"), n_file
                with open(plan[0], "wb") as f:
                    step = (1 << 30) // REC * REC
                    for o in range(0, n_file, step):
                        m = min(step, n_file - o)
                        f.write(device.synth_fastq(m, file_offset=o)[:m].cpu().numpy().tobytes())
        except Exception as e:      # noqa: BLE001 (disk full, ...)
            plan[0], plan[2] = None, f"{type(e).__name__}: {e}"
    dist.broadcast_object_list(plan, src=0)
    path, n_file, why = plan
    out = None
    if path is None:
        out = {"skipped": why}
    else:
        dev = 0 if args.single_device else local_rank
        n, dt, err = 0, 0.0, ""
        try:
            reader_count(lib, path, "fastq", (rank, world), dev)  # warm
        except Exception as e:      # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        dist.barrier()
        if not err:
            try:
                n, dt = reader_count(lib, path, "fastq", (rank, world), dev)
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        bad = torch.tensor([1 if err else 0], dtype=torch.int64, device=coll_dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=coll_dev))
        dist.all_reduce(bad)
        counts = [int(c.item()) for c in counts]
        # content (untimed): every rank pulls all four columns of ITS shard as DataChunks and folds every row — with the row's
        # index in the FILE, which is the sum of the shards' counts in front — into a digest; the shards' digests must add up
        # (mod 2^64) to the generator's digest of the whole file: the shards partition the rows, in file order, bit for bit
        dg, v_rows, wrong = 0, 0, 0
        if not int(bad.item()):
            try:
                v_rows, _, dg, wrong = reader_digest(lib, path, "fastq", 150, 0, (rank, world), dev, sum(counts[:rank]))
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
        parts = [None] * world
        dist.all_gather_object(parts, (dg, v_rows, wrong, err))
        if int(bad.item()) or any(p[3] for p in parts):
            out = {"skipped": "rank(s) failed: " + "; ".join(f"rank {i}: {p[3]}" for i, p in enumerate(parts) if p[3])}
        else:
            total_rows = sum(counts)
            verified = total_rows == n_file // REC and [p[1] for p in parts] == counts and not any(p[2] for p in parts)
            if rank == 0 and verified:
                from exon_duckdb_amd import abi, load_test_library
                want = int(load_test_library().exon_tf_expect_fastq150(abi.EXG_SYNTH_FASTQ_SEED, 0, n_file // REC, effective_cores()))
                verified = (sum(p[0] for p in parts) & (2 ** 64 - 1)) == want
            out = {"workload": f"COUNT(*) of one {n_file / 1e9:.1f} GB FASTQ-150 file in shared memory, {world} readers with shard_index = rank (exg_open), PCIe inclusive",
                   "algorithmic_bytes": n_file, "ms": float(t.item()) * 1e3, "GB/s": n_file / float(t.item()) / 1e9,
                   "records_per_s": total_rows / float(t.item()), "rows_per_shard": counts,
                   "verification": "COUNT(*) timed; an untimed pass pulls every shard's four columns as DataChunks and folds every row (with its "
                                   "index in the file) into a digest; the shards' digests must add up to the generator's for the whole file",
                   "verified": bool(verified)}
    dist.barrier()
    if rank == 0 and tmp:
        try:
            if plan[0] and os.path.exists(plan[0]):
                os.unlink(plan[0])
            os.rmdir(tmp)
        except OSError:
            pass
    return out


if __name__ == "__main__":
    main()
