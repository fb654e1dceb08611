"""One process per shard (the deployment shape of SURVEY §8 E1), run from a fresh shell — not from a process that has
touched the GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/shard_ranks.py
Each rank opens the same file with shard_index = RANK (device = LOCAL_RANK when there are that many GPUs, else 0),
reads its rows, and the ranks all_reduce the counts over gloo; rank 0 checks the union against the whole file."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from oracle import pyoracle            # test infrastructure: writes the input, provides the expected rows
from exon_duckdb_amd.reader import ShardReader

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
path = "/tmp/exg_shard_ranks.fastq"
n_rec = 200_000
if rank == 0:
    open(path, "wb").write(bytes(pyoracle.synth_fastq_ragged(n_rec)))
dist.barrier()
dev = int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.device_count() > int(os.environ.get("LOCAL_RANK", "0")) else 0
r = ShardReader(path, "fastq", shard_index=rank, shard_count=world, device=dev)
rows = r.rows()
r.close()
n = torch.tensor([len(rows)], dtype=torch.int64)
dist.all_reduce(n)
names = [None] * world
dist.all_gather_object(names, [row[0] for row in rows])
if rank == 0:
    exp = pyoracle.fastq_parse(open(path, "rb").read(), want_string_t=False)
    want = [exp.columns["name"].row(i) for i in range(exp.n_rows)]
    got = [x for part in names for x in part]
    assert int(n.item()) == n_rec == len(want) and got == want
    print(f"{world} ranks: {[len(p) for p in names]} rows, {int(n.item())} in all, names in file order: ok")
    os.unlink(path)
dist.destroy_process_group()
