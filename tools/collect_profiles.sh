#!/bin/bash
# Round profile collection, run on the GPU box from the repo root:
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02_a'
# Writes gpurun_out/<tag>_*; tools/pmc_summary.py then condenses them into profiles/.
set -u
TAG=${1:-r04_c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
# the bench line as the driver takes it (all configs, cpu baseline)
T0=$(date +%s)
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench.py (default flags): $(( $(date +%s) - T0 )) s"
cd /tmp
# headline kernel: per-kernel times of the same command (without the side legs), then the HBM counters in passes of their own
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt --output-format csv -- python3 $ROOT/bench.py --no-configs --no-cpu-baseline > $OUT/${TAG}_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/${TAG}_pmc_$c -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --launches-per-step 2 --no-configs --no-cpu-baseline > $OUT/${TAG}_pmc_$c.log 2>&1
done
# the configs legs (config 1 FASTA, config 3 VCF, config 4 BGZF inflate + CRC-32, end to end): every kernel of them
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_configs -o kt --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --launches-per-step 2 --no-cpu-baseline --gz-gb 2 --e2e-gb 2 > $OUT/${TAG}_kt_configs.log 2>&1
# records of other shapes (long reads, 36 bp reads, multi-sample VCF): the any-shape scan k_fused<.., 1>, the lean scan + redo run, k_*_far
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_shapes -o kt --output-format csv -- python3 $ROOT/tools/shapes_probe.py 4 > $OUT/${TAG}_kt_shapes.log 2>&1
# zstd through the reader (a 4 GB single frame, level 3, no Content_Checksum: three timed passes + the warm-up; the probe reads its knobs from the environment)
export ZST_GB=4 ZST_BATCHES=0 ZST_CHECK=0
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_zstd -o kt --output-format csv -- python3 $ROOT/tools/zstd_stream_probe.py > $OUT/${TAG}_kt_zstd.log 2>&1
unset ZST_GB ZST_BATCHES ZST_CHECK
# single-member gzip through the reader (chunked decode)
GZ_RECORDS=800000 GZ_ONLY_SINGLE=1 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_gzstream -o kt --output-format csv -- python3 $ROOT/tools/gz_probe.py > $OUT/${TAG}_kt_gzstream.log 2>&1
# FASTA (1 GB), VCF (5 GB), BGZF inflate alone: per-kernel times
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_fasta -o kt --output-format csv -- python3 $ROOT/tools/bench_fasta.py > $OUT/${TAG}_kt_fasta.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_vcf -o kt --output-format csv -- python3 $ROOT/tools/bench_vcf.py > $OUT/${TAG}_kt_vcf.log 2>&1
# read_fasta through the reader (round 6: uploads ahead, the joined sequences back by k_stream_to_host): a 2 GB file, COUNT(*) and all-columns drains
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_fasta_e2e -o kt --output-format csv -- python3 $ROOT/tools/fasta_e2e_probe.py > $OUT/${TAG}_kt_fasta_e2e.log 2>&1
# the nested VCF columns through the reader (round 6: exg_vcf_nested.hip): VCF-8 (2.08 GB, one all-columns drain) and cohort lines of 100 / 2 504
# samples with FORMAT GT (1 GB each, two drains); then the HBM counters of the VCF-8 drain, a pass each
export ONE_PASS=1
# (PROBE_COLUMNS=2: only `pos` is copied back, the nested kernels run alone — the profiler turns this process's big D2H copies into blit kernels that stretch whatever runs beside them)
PROBE_COLUMNS=2 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_vcf_nested -o kt --output-format csv -- python3 $ROOT/tools/vcf_nested_probe.py > $OUT/${TAG}_kt_vcf_nested.log 2>&1
SAMPLES=100 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_vcf_cohort100 -o kt --output-format csv -- python3 $ROOT/tools/vcf_cohort_probe.py > $OUT/${TAG}_kt_vcf_cohort100.log 2>&1
SAMPLES=2504 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_vcf_cohort2504 -o kt --output-format csv -- python3 $ROOT/tools/vcf_cohort_probe.py > $OUT/${TAG}_kt_vcf_cohort2504.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  PROBE_COLUMNS=2 rocprofv3 --kernel-trace --pmc $c -d $OUT/${TAG}_pmcn_$c -o pmc --output-format csv -- python3 $ROOT/tools/vcf_nested_probe.py > $OUT/${TAG}_pmcn_$c.log 2>&1
done
unset ONE_PASS
INFLATE_K=32 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_inflate -o kt --output-format csv -- python3 $ROOT/tools/bench_inflate.py > $OUT/${TAG}_kt_inflate.log 2>&1
cd $ROOT && bash tools/pmc_inflate.sh $TAG > $OUT/${TAG}_pmcinf.log 2>&1; cd /tmp
find $OUT -name "*kernel_stats.csv" -newer $OUT/${TAG}_bench.json | head
