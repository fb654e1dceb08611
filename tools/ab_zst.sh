#!/bin/bash
# A/B of library variants through the zstd reader probe: bash tools/ab_zst.sh v_a.so v_b.so   (ZST_GB, ZST_CHECK as the probe reads them)
for lib in "$@"; do
  cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
  for c in 1 0; do
    echo "$lib checksum=$c: $(ZST_CHECK=$c ZST_GB=${ZST_GB:-4} ZST_BATCHES=0 python tools/zstd_stream_probe.py 2>&1 | grep device_batch | tail -1)"
  done
  ZST_CHECK=0 ZST_GB=${ZST_GB:-4} ZST_BATCHES=0 EXG_TRACE=1 python tools/zstd_stream_probe.py 2>&1 | grep "producer: round" | tail -3
done
