#!/bin/bash
# kernel + memory-copy timeline of the BGZF all-columns drain (tools/gz_chunks_probe.py) -> gpurun_out/gzc_tl/*.csv
ROOT=$(pwd); export TMPDIR=/tmp GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8} GZC_GB=${GZC_GB:-2}
rm -rf $ROOT/gpurun_out/gzc_tl
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d $ROOT/gpurun_out/gzc_tl -o tl --output-format csv -- python3 $ROOT/tools/gz_chunks_probe.py > $ROOT/gpurun_out/gzc_tl.log 2>&1
tail -5 $ROOT/gpurun_out/gzc_tl.log; find $ROOT/gpurun_out/gzc_tl -name "*.csv" | xargs ls -la
