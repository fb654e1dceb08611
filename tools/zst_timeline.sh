#!/bin/bash
# kernel + memory-copy timeline of a zstd frame through the reader (tools/zstd_stream_probe.py: COUNT(*) x3, then with
# ZST_CHUNKS=1 all columns x3) -> gpurun_out/zst_tl/*.csv         gpurun -- 'bash tools/zst_timeline.sh'
ROOT=$(pwd); export TMPDIR=/tmp GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8} ZST_GB=${ZST_GB:-4} ZST_CHECK=${ZST_CHECK:-0} ZST_BATCHES=0 ZST_CHUNKS=${ZST_CHUNKS-1}
rm -rf $ROOT/gpurun_out/zst_tl; mkdir -p $ROOT/gpurun_out
cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $ROOT/gpurun_out/zst_tl -o tl --output-format csv -- python3 $ROOT/tools/zstd_stream_probe.py > $ROOT/gpurun_out/zst_tl.log 2>&1
grep -E "device_batch|all columns" $ROOT/gpurun_out/zst_tl.log; find $ROOT/gpurun_out/zst_tl -name "*.csv" | xargs ls -la
