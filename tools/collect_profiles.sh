#!/bin/bash
# Round profile collection, run on the GPU box from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01_d'
# Writes gpurun_out/<tag>_*; tools/pmc_summary.py then condenses them into profiles/.
set -u
TAG=${1:-r01_d}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/${TAG}_pmc_$c -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_$c.log 2>&1
done
ls -R $OUT | head -50
# inflate kernels (BGZF members; single-member stream through the reader)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_inflate -o kt --output-format csv -- python3 $ROOT/tools/bench_inflate.py > $OUT/${TAG}_kt_inflate.log 2>&1
GZ_RECORDS=800000 GZ_ONLY_SINGLE=1 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_gzstream -o kt --output-format csv -- python3 $ROOT/tools/gz_probe.py > $OUT/${TAG}_kt_gzstream.log 2>&1
ls $OUT/${TAG}_kt_inflate $OUT/${TAG}_kt_gzstream
