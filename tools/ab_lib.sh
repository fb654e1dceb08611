#!/bin/bash
# A/B of library variants inside ONE gpurun box: tools/ab_lib.sh "<command>" a.so b.so ...   (libs under exon_duckdb_amd/lib)
cmd=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
    echo "round $round $lib: $(timeout 600 bash -c "$cmd" 2>/dev/null | tail -1)"
  done
done
