#!/usr/bin/env python3
"""bench.py's record_shapes legs alone (run on the GPU box from the repo root): python tools/shapes_probe.py [GB [leg-name prefix]]"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exon_duckdb_amd import load_library
args = types.SimpleNamespace(shape_gb=float(sys.argv[1]) if len(sys.argv) > 1 else 2.0, shape_only=sys.argv[2] if len(sys.argv) > 2 else "")
out = bench.run_record_shapes(torch, load_library(), args)
for k, v in out.items():
    print(k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a not in ("workload", "algo", "first_batch_algo", "indexed_algo")}))
