#!/bin/bash
# A/B of library variants on the FASTA scan inside ONE gpurun box: tools/ab_fasta.sh a.so b.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
    echo "round $round $lib: $(timeout 300 python tools/bench_fasta.py 2>/dev/null | tail -1)"
  done
done
