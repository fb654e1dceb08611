#!/bin/bash
# SQ activity counters of k_inflate (run on the GPU box from the repo root): tools/pmc_inflate.sh <tag>
TAG=${1:-x}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_LDS"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  INFLATE_K=16 rocprofv3 --kernel-trace --pmc $set -d $OUT/${TAG}_pmcinf_$name -o pmc --output-format csv -- python3 $ROOT/tools/bench_inflate.py > $OUT/${TAG}_pmcinf_$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/${TAG}_pmcinf_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_inflate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, sum(v) / len(v), len(v))
PY
