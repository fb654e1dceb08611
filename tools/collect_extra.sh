#!/bin/bash
# kernel stats of the other scans (VCF config 3, FASTA 1 GB) + the VCF kernel's instruction counters
TAG=${1:-r01_h}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_vcf -o kt --output-format csv -- python3 $ROOT/tools/bench_vcf.py > $OUT/${TAG}_kt_vcf.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_fasta -o kt --output-format csv -- python3 $ROOT/tools/bench_fasta.py > $OUT/${TAG}_kt_fasta.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $OUT/${TAG}_pmc_vcf -o pmc --output-format csv -- python3 $ROOT/tools/bench_vcf.py > $OUT/${TAG}_pmc_vcf.log 2>&1
tail -12 $OUT/${TAG}_kt_vcf.log | grep -E "ms|GBps"
