#!/bin/bash
# A/B of prebuilt library variants inside ONE gpurun box (devices differ by up to 10 %): tools/ab_bench.sh a.so b.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
    r=$(timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))")
    echo "round $round $lib: $r"
  done
done
