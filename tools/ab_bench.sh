#!/bin/bash
# A/B of prebuilt library variants inside ONE gpurun box (devices differ by up to 10 %): tools/ab_bench.sh a.so b.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
    r=$(timeout 600 python bench.py --steps 10 --warmup 2 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/launch  frac %.4f  verified %s' % (d['config']['ms_per_launch'], d['roofline']['frac'], d['verified']))")
    echo "round $round $lib: $r"
  done
done
