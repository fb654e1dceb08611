#!/bin/bash
# SQ counters of one kernel over library builds, inside ONE box (run on the GPU box from the repo root):
#   tools/pmc_kernel.sh <tag> <kernel substring> "<command>" a.so b.so ...     (libs under exon_duckdb_amd/lib)
# Counter passes are separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md: never with other trace domains).
TAG=$1; KERNEL=$2; CMD=$3; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
# the program itself must follow `--` (no env / bash -c / taskset / *.py run directly: _profcmd.sh says why); relative script paths are resolved against the repo root
. "$ROOT/tools/_profcmd.sh"; profcmd_check "$CMD" || exit 2; CMD=$(profcmd_abs "$ROOT" "$CMD")
SETS=${PMC_SETS:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY|SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY|SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"}
for lib in "$@"; do
  cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
  IFS='|' read -ra arr <<< "$SETS"
  k=0
  for set in "${arr[@]}"; do
    k=$((k+1))
    d=$OUT/${TAG}_pmc_${lib%.so}_$k
    rm -rf $d
    (cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $d -o pmc --output-format csv -- $CMD > $d.log 2>&1)
  done
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/${TAG}_pmc_${lib%.so}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERNEL" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$lib", " ".join("%s=%.4g(n=%d)" % (k, sum(v) / len(v), len(v)) for k, v in sorted(acc.items())))
PY
done
