#!/usr/bin/env python3
"""exg_zstd_decode on .zst files (run on the GPU box): python tools/zstd_decode_file.py a.zst b.zst -> rc + size or the error"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import zstd_soak
for f in sys.argv[1:]:
    rc, out = zstd_soak.decode(open(f, "rb").read())
    print(f, rc, out if rc else len(out))
