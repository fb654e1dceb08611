#!/bin/bash
# the decoders' small results posted by a kernel vs copied (EXG_POST_BY_COPY=1), zstd through the reader:
#   gpurun -- 'bash tools/ab_post.sh'        ZST_GB=4 in ONE frame with its content checksum (bench.py's zstd leg) by default
export ZST_GB=${ZST_GB:-4} ZST_FRAME_MB=${ZST_FRAME_MB:-0} ZST_BATCHES=0 ZST_CHUNKS=1 GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
  echo "posted: $(timeout 600 python3 tools/zstd_stream_probe.py 2>&1 | grep -E 'device_batch|all columns' | tr '\n' '|')"
  echo "copied: $(EXG_POST_BY_COPY=1 timeout 600 python3 tools/zstd_stream_probe.py 2>&1 | grep -E 'device_batch|all columns' | tr '\n' '|')"
done
