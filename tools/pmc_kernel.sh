#!/bin/bash
# SQ activity + HBM counters of ONE kernel of a probe (run on the GPU box from the repo root):
#   bash tools/pmc_kernel.sh <tag> <kernel name substring> <python script> [its arguments]
# e.g. bash tools/pmc_kernel.sh r06_c_vcf100 "k_fused<exg::VcfFormat, 1>" tools/shapes_probe.py 4 vcf_multisample_100
# -> gpurun_out/<tag>_pmck_*/ and gpurun_out/<tag>_pmc_sq.csv (copy it into profiles/)
TAG=$1; KERNEL=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/${TAG}_pmck_$name -o pmc --output-format csv -- python3 $SCRIPT "$@" > $OUT/${TAG}_pmck_$name.log 2>&1
done
KERNEL="$KERNEL" python3 - <<PY
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/${TAG}_pmck_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if os.environ["KERNEL"] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/${TAG}_pmc_sq.csv", "w") as out:
    out.write("counter,average_per_dispatch,dispatches\n")
    for k, v in sorted(acc.items()):
        print(k, sum(v) / len(v), len(v))
        out.write(f"{k},{sum(v) / len(v)},{len(v)}\n")
PY
