"""Byte-range shards at the reader boundary (exg_open_args.shard_index / shard_count): the shards of a
file partition its rows, in file order, whatever the cut points hit — mid record, mid line, on a quality line that
starts with '@', inside the VCF header."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def whole(path, fmt):
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(path, fmt)
    rows = r.rows()
    r.close()
    return rows


def sharded(path, fmt, n, **kw):
    from exon_duckdb_amd.reader import ShardReader
    rows, counts = [], []
    for i in range(n):
        r = ShardReader(path, fmt, shard_index=i, shard_count=n, **kw)
        part = r.rows()
        r.close()
        c = ShardReader(path, fmt, shard_index=i, shard_count=n, **kw)
        counts.append(c.count())
        c.close()
        assert counts[-1] == len(part)
        rows.extend(part)
    return rows, counts


@pytest.mark.parametrize("n_shards", [2, 3, 8, 61])
def test_fastq_shards_partition_the_rows(gpu, oracle, tmp_path, n_shards):
    data = bytes(oracle.synth_fastq_ragged(30000))          # ragged: CRLF, missing descriptions, no final newline
    p = tmp_path / "ragged.fastq"
    p.write_bytes(data)
    want = whole(str(p), "fastq")
    assert len(want) == 30000
    got, counts = sharded(str(p), "fastq", n_shards)
    assert got == want
    assert sum(counts) == 30000 and min(counts) > 0


def test_fastq_150_quality_lines_that_start_with_at(gpu, oracle, tmp_path):
    # 1/41 of the quality lines of the synthetic FASTQ-150 start with '@': cuts land on them, the phase must hold
    data = bytes(oracle.synth_fastq(332 * 50000))
    p = tmp_path / "f150.fastq"
    p.write_bytes(data)
    want = whole(str(p), "fastq")
    for n in (7, 97):
        got, _ = sharded(str(p), "fastq", n)
        assert got == want


def test_shards_with_small_device_batches_and_small_halo(gpu, oracle, tmp_path, monkeypatch):
    data = bytes(oracle.synth_fastq_ragged(20000))
    p = tmp_path / "r.fastq"
    p.write_bytes(data)
    want = whole(str(p), "fastq")
    monkeypatch.setenv("EXG_SHARD_HALO", "4096")
    got, _ = sharded(str(p), "fastq", 5, device_batch_bytes=65536)
    assert got == want


@pytest.mark.parametrize("n_shards", [2, 5, 33])
def test_vcf_shards_partition_the_rows(gpu, oracle, tmp_path, n_shards):
    data = bytes(oracle.synth_vcf(20000))
    p = tmp_path / "s.vcf"
    p.write_bytes(data)
    want = whole(str(p), "vcf")
    assert len(want) == 20000
    got, counts = sharded(str(p), "vcf", n_shards)
    assert got == want and sum(counts) == 20000


def test_more_shards_than_records(gpu, golden_dir):
    want = whole(f"{golden_dir}/test.fastq", "fastq")
    got, counts = sharded(f"{golden_dir}/test.fastq", "fastq", 16)
    assert got == want and sum(counts) == 2
    want = whole(f"{golden_dir}/vcf/index.vcf", "vcf")
    got, _ = sharded(f"{golden_dir}/vcf/index.vcf", "vcf", 9)
    assert got == want and len(got) == 621


def _bgzf(data, block=65280):
    import struct
    import zlib
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" +
                   struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1) + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")   # + BGZF EOF marker


@pytest.mark.parametrize("n_shards", [2, 3, 8, 40])
@pytest.mark.parametrize("block", [65280, 5000])
def test_bgzf_fastq_shards_by_member_ranges(gpu, oracle, tmp_path, n_shards, block):
    # BGZF: the members are divided among the shards, each rank uploads and inflates only its own (+ a halo of
    # members in front); 5000-byte members put ~200 of them into the 1 MiB halo, 65280-byte ones ~17
    data = bytes(oracle.synth_fastq_ragged(25000))
    plain = tmp_path / "p.fastq"
    plain.write_bytes(data)
    gz = tmp_path / "p.fastq.gz"
    gz.write_bytes(_bgzf(data, block))
    want = whole(str(plain), "fastq")
    assert whole(str(gz), "fastq") == want
    got, counts = sharded(str(gz), "fastq", n_shards)
    assert got == want and sum(counts) == 25000


def test_bgzf_more_shards_than_members(gpu, golden_dir, tmp_path):
    data = open(f"{golden_dir}/test.fastq", "rb").read()
    gz = tmp_path / "t.fastq.gz"
    gz.write_bytes(_bgzf(data))
    got, counts = sharded(str(gz), "fastq", 7)
    assert got == whole(f"{golden_dir}/test.fastq", "fastq") and sum(counts) == 2


def test_unshardable_inputs_fail_loudly(gpu, golden_dir):
    from exon_duckdb_amd._lib import ExgError
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(f"{golden_dir}/test.fasta.gz", "fasta", shard_index=0, shard_count=2)           # plain gzip: no member sizes
    with pytest.raises(ExgError, match="BGZF"):
        r.rows()
    r.close()
    r = ShardReader(f"{golden_dir}/test.fastq.gz", "fastq", shard_index=0, shard_count=2)   # plain gzip: no member sizes
    with pytest.raises(ExgError, match="BGZF"):
        r.rows()
    r.close()
    with pytest.raises(ExgError):
        ShardReader(f"{golden_dir}/test.fastq", "fastq", shard_index=2, shard_count=2)


@pytest.mark.parametrize("n_shards", [2, 3, 7, 16])
def test_record_longer_than_the_halo_across_a_cut(gpu, tmp_path, n_shards):
    """A record longer than the 1 MiB halo that crosses a cut (an ultra-long read; with 16 shards it spans whole shards): it
    belongs to the shard its last line ends in, and that shard looks as far back as it has to (EXG_RF_HEAD_UNRESOLVED ->
    the halo grows eightfold, the batch is scanned again).  The shards partition the rows of the unsharded scan — text and
    BGZF (whose halo is made of members: they are chosen again) — where round 2 raised an error."""
    short = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(1000))
    long_read = b"@long some description\n" + b"ACGT" * 750_000 + b"\n+\n" + b"I" * 3_000_000 + b"\n"
    data = short + long_read + short + b"@long2\n" + b"C" * 1_500_000 + b"\n+\n" + b"#" * 1_500_000 + b"\n" + short
    p = tmp_path / "long.fastq"
    p.write_bytes(data)
    want = whole(str(p), "fastq")
    assert len(want) == 3002
    got, counts = sharded(str(p), "fastq", n_shards)
    assert got == want
    gz = tmp_path / "long.fastq.gz"
    gz.write_bytes(_bgzf(data))
    got, counts = sharded(str(gz), "fastq", n_shards)
    assert got == want


def test_vcf_line_longer_than_the_halo_across_a_cut(gpu, oracle, tmp_path):
    """a very wide multi-sample line (3 MB of samples) across the cuts of 2 .. 5 shards"""
    body = bytes(oracle.synth_vcf(3000))
    lines = body.split(b"\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b"#CHROM"))
    wide = b"1\t999\t.\tA\tC\t.\t.\tDP=1\tGT\t" + b"\t".join(b"0/1" for _ in range(750_000))
    mid = k + 1 + (len(lines) - k) // 2
    data = b"\n".join(lines[:mid] + [wide] + lines[mid:])
    p = tmp_path / "wide.vcf"
    p.write_bytes(data)
    want = whole(str(p), "vcf")
    assert len(want) == 3001
    for n in (2, 3, 5):
        got, _ = sharded(str(p), "vcf", n)
        assert [r[:2] for r in got] == [r[:2] for r in want] and len(got) == 3001


@pytest.mark.parametrize("n_shards", [1, 3, 7])
def test_cohort_vcf_shards_under_the_indexed_scan(gpu, oracle, tmp_path, monkeypatch, n_shards):
    """lines of 3 - 14 kB in batches of 1 MiB: behind its first batches a shard's reader leaves the rows to a kernel of their own
    (EXG_ALGO_FUSED_INDEX, round 5) — halo lines in front of a shard's first own line, lines across batch ends, CRLF, '.' QUAL, a
    last line without a newline; the shards' rows are the file's rows in order, with and without the switch"""
    from exon_duckdb_amd import abi
    from exon_duckdb_amd.reader import ShardReader
    rng = np.random.default_rng(21)
    hdr = b"##fileformat=VCFv4.2\n##contig=<ID=chr1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts\n"
    lines = []
    for k in range(1500):
        ns = int(rng.integers(600, 2800))
        eol = b"\r\n" if k % 11 == 0 else b"\n"
        lines.append(b"chr1\t%d\trs%d\t%s\tC\t%s\tPASS\tDP=%d;AF=0.25\tGT:DP" % (1000 + k, k, b"A" * (1 + k % 20), b"." if k % 4 == 0 else b"%d.5" % (k % 97), k)
                     + b"\t0/1:7" * ns + eol)
    data = hdr + b"".join(lines)
    data = data[:-1]  # no last newline
    p = tmp_path / "cohort.vcf"
    p.write_bytes(data)
    t = oracle.vcf_parse(data, want_string_t=False)
    assert t.error_code == 0 and t.n_rows == 1500
    want = list(zip(t.columns["chrom"].to_list(), [int(x) for x in t.extra["pos"]], t.columns["ref"].to_list(),
                    [float(q) if v else None for q, v in zip(t.extra["qual"], t.extra["qual_valid"])]))
    monkeypatch.setenv("EXG_SHARD_HALO", "4096")  # (smaller than a line: the shard looks further back and scans again)
    for no_index in (False, True):
        if no_index:
            monkeypatch.setenv("EXG_NO_VCF_INDEX", "1")
        got, algos = [], set()
        for i in range(n_shards):
            r = ShardReader(str(p), "vcf", shard_index=i, shard_count=n_shards, device_batch_bytes=1 << 20, columns=[0, 1, 3, 5])
            got.extend(r.rows())
            algos.add(r.stats()["scan_algo"])
            r.close()
        assert got == want
        assert (abi.EXG_ALGO_FUSED_INDEX in algos) == (not no_index)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EXG_SHARD_FUZZ", "12"))))
def test_random_shard_geometry(gpu, oracle, tmp_path, monkeypatch, seed):
    # random record counts, shard counts, halo sizes and device batch sizes; text and BGZF (random member sizes)
    rng = np.random.default_rng(500 + seed)
    n_rec = int(rng.integers(1, 6000))
    fmt = ("vcf", "fastq", "fastq", "fasta")[seed % 4]
    data = bytes(oracle.synth_fastq_ragged(n_rec, seed=900 + seed)) if fmt == "fastq" else \
        bytes(oracle.synth_vcf(n_rec)) if fmt == "vcf" else bytes(oracle.synth_fasta(n_rec, seed=900 + seed))
    p = tmp_path / f"g.{fmt}"
    p.write_bytes(data)
    want = whole(str(p), fmt)
    assert len(want) == n_rec
    monkeypatch.setenv("EXG_SHARD_HALO", str(int(rng.choice([2048, 4096, 65536, 1 << 20]))))
    n_shards = int(rng.integers(2, 50))
    batch = int(rng.choice([0, 16384, 65536, 1 << 20]))
    got, counts = sharded(str(p), fmt, n_shards, device_batch_bytes=batch)
    assert got == want and sum(counts) == n_rec, (seed, fmt, n_rec, n_shards, batch)
    gz = tmp_path / f"g.{fmt}.gz"
    gz.write_bytes(_bgzf(data, int(rng.integers(300, 65280))))
    got, counts = sharded(str(gz), fmt, n_shards, device_batch_bytes=batch)
    assert got == want and sum(counts) == n_rec, (seed, "bgzf", fmt, n_rec, n_shards, batch)


@pytest.mark.parametrize("n_shards", [2, 7, 40])
def test_bgzip_vcf_shards_with_a_header_of_many_members(gpu, oracle, golden_dir, tmp_path, n_shards):
    # the reference's own bgzip fixture, and a VCF whose header alone fills ~90 BGZF members (every rank inflates the
    # leading members until the '#' lines end; a halo that would begin among them is taken from the start of the file)
    want = whole(f"{golden_dir}/vcf/index.vcf", "vcf")
    got, counts = sharded(f"{golden_dir}/vcf/index.vcf.gz", "vcf", n_shards)
    assert got == want and sum(counts) == 621
    body = bytes(oracle.synth_vcf(5000))
    lines = body.split(b"\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b"#CHROM"))
    filler = b"".join(b"##contig=<ID=scaffold_%07d,length=%d>\n" % (i, 1000 + i) for i in range(120_000))
    data = b"\n".join(lines[:k]) + b"\n" + filler + b"\n".join(lines[k:])
    plain = tmp_path / "h.vcf"
    plain.write_bytes(data)
    gz = tmp_path / "h.vcf.gz"
    gz.write_bytes(_bgzf(data))
    want = whole(str(plain), "vcf")
    assert len(want) == 5000
    got, counts = sharded(str(gz), "vcf", n_shards)
    assert got == want and sum(counts) == 5000


@pytest.mark.parametrize("n_shards", [2, 5, 33])
def test_fasta_shards_are_runs_of_whole_records(gpu, oracle, golden_dir, tmp_path, n_shards):
    # a record belongs to the shard in whose bytes its '>' line begins; one 3 MB sequence in the middle spans several
    # shards' byte ranges and is still one row of one shard
    recs = bytes(oracle.synth_fasta(1500))
    big = b">chrBig a long one\n" + b"\n".join(b"ACGT" * 15 for _ in range(50_000)) + b"\n"
    data = recs + big + recs.replace(b">", b">second_")
    p = tmp_path / "s.fasta"
    p.write_bytes(data)
    want = whole(str(p), "fasta")
    assert len(want) == 3001
    got, counts = sharded(str(p), "fasta", n_shards)
    assert got == want and sum(counts) == 3001
    want = whole(f"{golden_dir}/test.fasta", "fasta")
    got, _ = sharded(f"{golden_dir}/test.fasta", "fasta", n_shards)
    assert got == want and len(got) == 2


def test_open_on_one_thread_scan_on_another(gpu, tmp_path, oracle):
    """HIP's current device is per thread: every reader entry point switches to the reader's device itself (DuckDB binds
    on one thread and scans on others)."""
    import threading
    from exon_duckdb_amd.reader import ShardReader
    data = bytes(oracle.synth_fastq(332 * 5000))
    p = tmp_path / "t.fastq"
    p.write_bytes(data)
    want = ShardReader(str(p), "fastq").rows()
    box = {}

    def opener():
        box["r"] = ShardReader(str(p), "fastq")

    def scanner():
        box["rows"] = box["r"].rows()

    def closer():
        box["r"].close()

    for fn in (opener, scanner, closer):
        t = threading.Thread(target=fn)
        t.start()
        t.join()
    assert box["rows"] == want and len(want) == 5000
    # a device that does not exist is an argument error, not a crash
    from exon_duckdb_amd import ExgError
    with pytest.raises(ExgError):
        ShardReader(str(p), "fastq", device=63)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_bench_two_ranks_on_one_gpu(gpu):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), here with two ranks sharing
    cuda:0 over gloo: the byte-range shards, the phase all_gather, the COUNT(*) all_reduce, the closed-form verification of
    both shards and the reader-level leg must produce one JSON line with the whole-job numbers."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--launches-per-step", "2", "--gb", "0.5", "--e2e-gb", "0.25", "--backend", "gloo", "--single-device"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["verified"] is True
    assert j["config"]["file_bytes"] == int(2 * 0.5e9) // 332 * 332
    assert abs(j["value"] - (j["config"]["file_bytes"] // 332) * 2 * 2 / (j["ms_per_step"] * 2 / 1e3)) < 1e-3 * j["value"]
    rs = j["reader_sharded"]
    assert rs.get("verified") is True, rs


def test_bench_launches_its_own_ranks(gpu):
    """`python3 bench.py --gpus 2 ...` with NO launcher (what the driver's 1-GPU command line looks like with another N): the
    script starts the two ranks itself, as a child torch.distributed.run, and its one JSON line says n_gpus = 2; the sharded
    reader leg is verified by digest (every row of every shard, with its index in the file)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "2",
           "--warmup", "1", "--launches-per-step", "2", "--gb", "0.5", "--e2e-gb", "0.25"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["verified"] is True
    rs = j["reader_sharded"]
    assert rs.get("verified") is True and len(rs["rows_per_shard"]) == 2 and "digest" in rs["verification"], rs


def test_bench_eight_ranks_on_one_gpu(gpu):
    """BASELINE config 5's process layout on a one-GPU box: `python3 bench.py --gpus 8` with no launcher starts EIGHT ranks (eight
    processes' pinned / device pools and ports on one box), each scans its byte-range shard of the 8-shard file (middle shards:
    halo, no BOF / EOF, guessed phase), the 8-way all_gather verifies the phases, the all_reduce gives COUNT(*), and the
    reader-level leg has eight readers of ONE shared file whose digests must add up to the generator's.  gloo + --single-device:
    the collectives' world size is printed in the line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--single-device", "--steps", "2",
           "--warmup", "1", "--launches-per-step", "2", "--gb", "0.25", "--e2e-gb", "0.1"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=840)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["verified"] is True and j["scaling"] == "weak"
    assert j["collectives"]["world_size"] == 8 and j["collectives"]["backend"] == "gloo"
    assert j["config"]["file_bytes"] == int(8 * 0.25e9) // 332 * 332
    rs = j["reader_sharded"]
    assert rs.get("verified") is True and len(rs["rows_per_shard"]) == 8 and sum(rs["rows_per_shard"]) * 332 == rs["algorithmic_bytes"], rs


def test_rccl_initialises_and_reduces_on_this_image(gpu):
    """bench.py's N > 1 collectives run over RCCL (torch's "nccl" backend).  A one-GPU box cannot run two ranks on it, but it
    can prove that the library loads, a communicator comes up and the three collectives the bench uses complete on the device."""
    import os
    import subprocess
    import sys
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '%d')\n"
        "os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "t = torch.tensor([41], dtype=torch.int64, device='cuda'); dist.all_reduce(t)\n"
        "g = [torch.zeros(1, dtype=torch.int64, device='cuda')]; dist.all_gather(g, t + 1)\n"
        "m = torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(m, op=dist.ReduceOp.MAX)\n"
        "dist.barrier(); torch.cuda.synchronize()\n"
        "assert int(t.item()) == 41 and int(g[0].item()) == 42 and float(m.item()) == 1.5\n"
        "dist.destroy_process_group(); print('rccl ok')\n") % _free_port()
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    assert res.returncode == 0 and "rccl ok" in res.stdout, res.stdout[-1000:] + res.stderr[-3000:]


# ---- shard_count = 0: one consumer-facing stream that fans out over stripes by itself (exg_rd_fanout.hpp) ---------------

def _auto(path, fmt, **kw):
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(path, fmt, shard_count=0, **kw)
    rows = r.rows()
    r.close()
    c = ShardReader(path, fmt, shard_count=0, **kw)
    n = c.count()
    c.close()
    assert n == len(rows)
    return rows


@pytest.mark.parametrize("workers", [1, 3, 8])
def test_fan_out_inside_one_reader(gpu, oracle, tmp_path, monkeypatch, workers):
    """exg_open with shard_count = 0 plans stripes (forced here: EXON_GPU_SHARDS stripes on the one device of the test box,
    EXG_FANOUT_WORKERS threads — on an 8-GPU node: a multiple of 8 stripes of ~1 GiB, one worker and one device each) and
    hands their batches out in file order: the rows of the unsharded scan, for every format the shards support, with small
    device batches so that every stripe holds several; COUNT(*) sums the stripes."""
    fq = bytes(oracle.synth_fastq_ragged(30000))
    vcf = bytes(oracle.synth_vcf(20000))
    fa = bytes(oracle.synth_fasta(3000, seed=77))
    files = {"p.fastq": (fq, "fastq"), "p.vcf": (vcf, "vcf"), "p.fasta": (fa, "fasta"), "b.fastq.gz": (_bgzf(fq, 20000), "fastq"),
             "b.vcf.gz": (_bgzf(vcf), "vcf")}
    want = {}
    for name, (blob, fmt) in files.items():
        (tmp_path / name).write_bytes(blob)
        want[name] = whole(str(tmp_path / name), fmt)
    monkeypatch.setenv("EXON_GPU_SHARDS", "7")
    monkeypatch.setenv("EXG_FANOUT_WORKERS", str(workers))
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(256 << 10))
    for name, (blob, fmt) in files.items():
        assert _auto(str(tmp_path / name), fmt) == want[name], name
    # a pushed-down filter runs in every stripe
    got = _auto(str(tmp_path / "p.vcf"), "vcf", filters="chrom='7'")
    assert got == [r for r in want["p.vcf"] if r[0] == b"7"] and got
    # an input that cannot be sharded (plain gzip) is simply read by the reader itself
    import gzip
    (tmp_path / "one.fastq.gz").write_bytes(gzip.compress(fq, 1, mtime=0))
    assert _auto(str(tmp_path / "one.fastq.gz"), "fastq") == want["p.fastq"]


def test_fan_out_reports_a_parse_error_behind_the_rows_in_front_of_it(gpu, oracle, tmp_path, monkeypatch):
    from exon_duckdb_amd._lib import ExgError
    from exon_duckdb_amd.reader import ShardReader
    from exon_duckdb_amd.table_function import Chunk
    import ctypes as C
    fq = bytearray(oracle.synth_fastq(332 * 20000))
    fq[332 * 15000] = ord("X")                      # record 15000 does not start with '@'
    p = tmp_path / "bad.fastq"
    p.write_bytes(bytes(fq))
    monkeypatch.setenv("EXON_GPU_SHARDS", "5")
    monkeypatch.setenv("EXG_FANOUT_WORKERS", "3")
    r = ShardReader(str(p), "fastq", shard_count=0)
    seen, rc = 0, 0
    while True:
        ch = Chunk()
        rc = r._l.exg_next_chunk(r._r, C.byref(ch))
        if rc != 0 or ch.n_rows == 0:
            break
        seen += int(ch.n_rows)
        r._l.exg_release_chunk(r._r, C.byref(ch))
    assert rc != 0 and seen == 15000
    r.close()
    with pytest.raises(ExgError):
        ShardReader(str(p), "fastq", shard_count=0).count()


# ---- shards of compressed inputs beyond BGZF FASTQ / VCF: bgzip FASTA by members, zstd by frames -----------------------------

@pytest.mark.parametrize("n_shards", [2, 5, 23])
@pytest.mark.parametrize("block", [65280, 3000])
def test_bgzip_fasta_shards(gpu, oracle, tmp_path, monkeypatch, n_shards, block):
    """A record belongs to the shard in whose members' bytes its '>' line begins: the shard's stream starts one member in
    front of its own (does their first byte begin a line?), its first record start is searched on the device, and it reads
    on behind its members until the next record start (the decoder tells where its own members end: mark 1).  Records of
    3 .. 50 lines, some longer than several members; more shards than records at 23 x 65280."""
    data = bytes(oracle.synth_fasta(400, seed=21))
    data += b">long one\n" + b"".join(b"ACGTACGTAC" * 6 + b"\n" for _ in range(9000)) + bytes(oracle.synth_fasta(50, seed=22))
    p = tmp_path / "s.fasta"
    p.write_bytes(data)
    want = whole(str(p), "fasta")
    gz = tmp_path / "s.fasta.gz"
    gz.write_bytes(_bgzf(data, block))
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(128 << 10))
    got, counts = sharded(str(gz), "fasta", n_shards)
    assert got == want and sum(counts) == len(want)


def _zst_frames(data, cuts, **kw):
    from zstd_util import compress, skippable
    out = []
    for i in range(len(cuts) - 1):
        out.append(compress(data[cuts[i]:cuts[i + 1]], 3, i % 2 == 0, content_size=kw.get("content_size", True) and i % 3 != 1))
        if i == 1:
            out.append(skippable(b"x" * 37))
    return b"".join(out)


@pytest.mark.parametrize("n_shards", [2, 3, 9])
def test_zstd_shards_by_frames(gpu, oracle, tmp_path, monkeypatch, n_shards):
    """A multi-frame zstd file (pzstd, the seekable format) is sharded by frames: a frame belongs to the shard in whose bytes
    it begins; the stream of a shard begins with a halo of frames in front of its own, and the decoder tells where its own
    bytes begin (some frames do not state their content size).  FASTQ (cuts mid record, mid line), VCF (every rank reads the
    header from the first frames), FASTA (the shard reads on to the next record start); 9 shards over 8 frames: empty ones."""
    import random
    rng = random.Random(n_shards)
    fq = bytes(oracle.synth_fastq_ragged(20000))
    vcf = bytes(oracle.synth_vcf(15000))
    fa = bytes(oracle.synth_fasta(2500, seed=4))
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(256 << 10))
    monkeypatch.setenv("EXG_SHARD_HALO", "65536")
    for name, data, fmt in (("m.fastq", fq, "fastq"), ("m.vcf", vcf, "vcf"), ("m.fasta", fa, "fasta")):
        cuts = sorted({0, len(data)} | {rng.randrange(1, len(data)) for _ in range(7)})
        (tmp_path / name).write_bytes(data)
        want = whole(str(tmp_path / name), fmt)
        z = tmp_path / (name + ".zst")
        z.write_bytes(_zst_frames(data, cuts))
        assert whole(str(z), fmt) == want
        got, counts = sharded(str(z), fmt, n_shards)
        assert got == want and sum(counts) == len(want), (name, n_shards)


def test_fan_out_over_compressed_fasta_and_zstd(gpu, oracle, tmp_path, monkeypatch):
    from exon_duckdb_amd.reader import ShardReader
    fa = bytes(oracle.synth_fasta(3000, seed=8))
    fq = bytes(oracle.synth_fastq(332 * 20000))
    (tmp_path / "a.fasta").write_bytes(fa)
    (tmp_path / "a.fastq").write_bytes(fq)
    (tmp_path / "a.fasta.gz").write_bytes(_bgzf(fa, 20000))
    (tmp_path / "a.fastq.zst").write_bytes(_zst_frames(fq, list(range(0, len(fq), 700_001)) + [len(fq)]))
    monkeypatch.setenv("EXON_GPU_SHARDS", "5")
    monkeypatch.setenv("EXG_FANOUT_WORKERS", "3")
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(256 << 10))
    assert _auto(str(tmp_path / "a.fasta.gz"), "fasta") == whole(str(tmp_path / "a.fasta"), "fasta")
    assert _auto(str(tmp_path / "a.fastq.zst"), "fastq") == whole(str(tmp_path / "a.fastq"), "fastq")
