#!/bin/bash
# the zstd resolve: inner launches + in-order groups of ~sqrt(n) chunks against round 4's groups of 512 KiB
#   gpurun -- 'bash tools/ab_resolve.sh'     ZST_GB=4 in ONE frame without a checksum by default
export ZST_GB=${ZST_GB:-4} ZST_FRAME_MB=${ZST_FRAME_MB:-0} ZST_CHECK=${ZST_CHECK:-0} ZST_BATCHES=0 ZST_CHUNKS=${ZST_CHUNKS-1} GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
  for v in ${AB_VARIANTS:-"EXG_ZSTD_RESOLVE_INNER=1" "EXG_ZSTD_RESOLVE_INNER=0" "EXG_ZSTD_GROUP_CHUNKS=32" "EXG_ZSTD_GROUP_CHUNKS=16"}; do
    echo "$v: $(env $v timeout 600 python3 tools/zstd_stream_probe.py 2>&1 | grep -E 'device_batch|all columns|Error|error' | tr '\n' '|')"
  done
done
