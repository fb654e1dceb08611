#!/bin/bash
# kernel + memory-copy timeline of config 4 (tools/c4_probe.py: COUNT(*) x4, all columns x3) -> gpurun_out/c4_tl/*.csv
#   gpurun -- 'bash tools/c4_timeline.sh'       C4_GB=3 by default
ROOT=$(pwd); export TMPDIR=/tmp GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8} C4_GB=${C4_GB:-3}
rm -rf $ROOT/gpurun_out/c4_tl; mkdir -p $ROOT/gpurun_out
timeout 300 python3 tools/c4_probe.py build || exit 1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $ROOT/gpurun_out/c4_tl -o tl --output-format csv -- python3 $ROOT/tools/c4_probe.py run > $ROOT/gpurun_out/c4_tl.log 2>&1
tail -3 $ROOT/gpurun_out/c4_tl.log; find $ROOT/gpurun_out/c4_tl -name "*.csv" | xargs ls -la
rm -f /dev/shm/exg_c4.fastq.gz
