"""N > 1 host path on CPU: world_size-2 gloo processes shard one synthetic FASTQ by byte ranges, exchange
newline counts (all_gather), verify the guessed phase, and all_reduce COUNT(*).  The per-shard "scan"
is the oracle restricted to the records that END in the shard (the ownership rule of the device scan),
so the test pins the protocol bench.py runs on GPUs: plan_shards -> scan -> all_gather -> verify ->
all_reduce."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_records, ragged, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from exon_duckdb_amd import sharding
    from oracle import pyoracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = pyoracle.synth_fastq_ragged(n_records) if ragged else pyoracle.synth_fastq(332 * n_records)
        nl = np.flatnonzero(data == 10)
        sh = sharding.plan_shards(len(data), world, halo=2048)[rank]
        # --- what the device scan returns for this shard: '\n' count in [start, end) (+ virtual EOF line)
        local_lines = int(((nl >= sh.start) & (nl < sh.end)).sum())
        if sh.is_last and len(data) and data[-1] != 10:
            local_lines += 1
        # --- phase guess from the shard's own bytes (what exg_fastq_guess_phase does)
        pos = 0 if sh.start == 0 else int(nl[np.searchsorted(nl, sh.start - 1)]) + 1
        starts = [pos]
        for _ in range(31):
            k = np.searchsorted(nl, starts[-1])
            if k >= len(nl) or nl[k] + 1 >= len(data):
                break
            starts.append(int(nl[k]) + 1)
        ok = [p for p in range(4) if all((data[s] == ord("@")) if (p + i) % 4 == 0 else (data[s] == ord("+")) if (p + i) % 4 == 2 else True
                                         for i, s in enumerate(starts))]
        lines_before_first_full_line = int((nl < pos).sum())
        guess_first_line = ok[0] if len(ok) == 1 and len(starts) >= 8 else None
        # --- exchange (8 bytes per rank) and verify
        counts = sharding.all_gather_int(local_lines)
        fli = sharding.first_line_index_from_counts(counts, rank)
        assert fli == int((nl < sh.start).sum())
        guess_at_start = None if guess_first_line is None else (guess_first_line - (lines_before_first_full_line - fli)) % 4
        assert sharding.phase_is_consistent(guess_at_start, fli), (rank, guess_at_start, fli)
        # --- records owned by the shard: quality line (line 4r+3) ends inside [start, end)
        exp = pyoracle.fastq_parse(data)
        T = len(nl) + (1 if len(data) and data[-1] != 10 else 0)
        ends = np.concatenate([nl, [len(data)]])[:T]
        q_ends = ends[3::4]
        owned = int(((q_ends >= sh.start) & ((q_ends < sh.end) | ((q_ends == len(data)) & sh.is_last))).sum())
        total = sharding.all_reduce_sum(owned)
        assert total == exp.n_rows == n_records
        q.put((rank, owned, fli))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_sharded_count(ragged):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 4000, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got[0][1] + got[1][1] == 4000
    assert got[0][2] == 0 and got[1][2] > 0


def test_eight_rank_sharded_count():
    """BASELINE config 5's collective shape on CPU: EIGHT gloo ranks (one per GPU of the node), six of them middle shards — no
    BOF, no EOF, a halo in front, a phase guessed from their own bytes — the 8-way all_gather of the newline counts verifies
    every guess and the all_reduce gives COUNT(*)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n = 16000
    procs = [ctx.Process(target=_worker, args=(r, 8, port, n, False, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(8))
    assert sum(g[1] for g in got) == n and [g[0] for g in got] == list(range(8))
    assert got[0][2] == 0 and all(a[2] < b[2] for a, b in zip(got[:-1], got[1:]))   # first line index of every shard: ascending


def test_plan_shards_properties():
    sys.path.insert(0, ROOT)
    from exon_duckdb_amd import sharding

    for n, w in [(10**10, 8), (332 * 1000, 3), (5000, 2), (100, 1)]:
        sh = sharding.plan_shards(n, w)
        assert sh[0].start == 0 and sh[-1].end == n and sh[0].halo == 0
        for a, b in zip(sh[:-1], sh[1:]):
            assert a.end == b.start and b.start % 16 == 0 and b.halo % 16 == 0 and b.halo <= b.start
        assert sum(s.end - s.start for s in sh) == n
