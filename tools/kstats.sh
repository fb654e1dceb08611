#!/bin/bash
# per-kernel time of a command over library builds inside ONE box: tools/kstats.sh <tag> "<command>" a.so b.so ...
TAG=$1; CMD=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
# the program itself must follow `--` (no env / bash -c / taskset / *.py run directly: _profcmd.sh says why); relative script paths are resolved against the repo root
. "$ROOT/tools/_profcmd.sh"; profcmd_check "$CMD" || exit 2; CMD=$(profcmd_abs "$ROOT" "$CMD")
for lib in "$@"; do
  cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
  d=$OUT/${TAG}_kt_${lib%.so}; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o kt --output-format csv -- $CMD > $d.log 2>&1)
  echo "== $lib"; f=$(find $d -name "*kernel_stats.csv" | head -1); head -${KSTATS_TOP:-8} $f | cut -d, -f1-4,7-8 | cut -c1-200
done
