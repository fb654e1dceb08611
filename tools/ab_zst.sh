#!/bin/bash
# A/B of library variants through the zstd reader probe: bash tools/ab_zst.sh v_a.so v_b.so   (ZST_GB as the probe reads it)
# per variant: the probe's best of three with and without the content checksum, and the producer's per-round decode times
for round in 1 2; do
for lib in "$@"; do
  cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
  for c in 1 0; do
    ZST_CHECK=$c ZST_GB=${ZST_GB:-4} ZST_BATCHES=0 EXG_TRACE=1 python tools/zstd_stream_probe.py > /tmp/z.log 2>&1
    echo "$round $lib checksum=$c: $(grep device_batch /tmp/z.log | tail -1 | cut -d: -f2 | cut -c1-36) | rounds: $(grep 'producer: round' /tmp/z.log | tail -4 | sed 's/.*compressed: //' | tr '\n' ' ')"
  done
done
done
