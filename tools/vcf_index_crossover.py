"""EXG_ALGO_FUSED_FULL against EXG_ALGO_FUSED_INDEX over line widths (samples per line): where the reader's switch belongs
(exg_rd_batch.cpp: an average of 2 KiB a line).  python tools/vcf_index_crossover.py [GB]   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exon_duckdb_amd import abi, device, load_library
from exon_duckdb_amd.testing import shapes

load_library()
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
BASE = 0x7E0000000000
for ns in (100, 200, 300, 400, 500, 650, 800, 1200, 2504):
    n_lines = max(2000, 40_000_000 // (ns * 4 + 90))
    hdr, block, e = shapes.vcf_multisample_block(n_lines, ns, seed=ns)
    reps = max(1, int(gb * 1e9) // len(block))
    n = len(hdr) + reps * len(block)
    d = torch.zeros((n + 15) // 16 * 16 + 64, dtype=torch.uint8, device="cuda")
    d[:len(hdr)].copy_(torch.frombuffer(bytearray(hdr), dtype=torch.uint8))
    d[len(hdr):n] = torch.frombuffer(bytearray(block), dtype=torch.uint8).cuda().repeat(reps)
    scan = device.VcfScan(n, capacity_records=n_lines * reps + 16)
    kw = dict(n_bytes=n, lead=len(hdr), payload_base=BASE)
    out = []
    for algo in (abi.EXG_ALGO_FUSED_FULL, abi.EXG_ALGO_FUSED_INDEX):
        ms, _ = bench.timed_launches(torch, lambda: scan.launch(d, algo=algo, **kw), 5, warm=1)
        r = scan.fetch()
        assert r.error_code == 0 and int(r.n_records) == n_lines * reps
        out.append(n / (ms * 1e-3) / 1e9)
    print(f"{ns:5d} samples, {len(block) // n_lines:6d} B a line: rows inside {out[0]:7.0f} GB/s | indexed {out[1]:7.0f} GB/s", flush=True)
    del scan, d
    torch.cuda.empty_cache()
