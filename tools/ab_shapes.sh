#!/bin/bash
# A/B of library variants on the record-shape legs (tools/shapes_probe.py): [LEGS=vcf] bash tools/ab_shapes.sh v_a.so v_b.so
for round in 1 2; do
  for lib in "$@"; do
    cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
    echo "round $round $lib: $(timeout 600 python tools/shapes_probe.py 3 $LEGS 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if ' ' not in l: continue
    k,v=l.split(' ',1)
    try: d=json.loads(v)
    except Exception: continue
    print(k, d['GB/s'], d.get('indexed_GB/s'), d['first_batch_GB/s'], d['verified'], end=' | ')")"
  done
done
