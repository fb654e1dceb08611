#!/bin/bash
# A/B of library variants on the VCF scan inside ONE gpurun box: tools/ab_vcf.sh a.so b.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
    echo "round $round $lib: $(timeout 300 python tools/bench_vcf.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('all %.3f ms  chrom,pos %.3f ms' % (d['all_columns']['ms'], d['chrom_pos_only']['ms']))")"
  done
done
