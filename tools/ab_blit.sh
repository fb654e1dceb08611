#!/bin/bash
# the HIP runtime's copy-engine knobs under the zstd all-columns drain (its 1 GiB mirrors leave as copy KERNELS):
#   gpurun -- 'bash tools/ab_blit.sh'
export ZST_GB=${ZST_GB:-4} ZST_CHECK=0 ZST_BATCHES=0 ZST_CHUNKS=1 GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
  for v in ${AB_VARIANTS:-"EXG_X=0" "DEBUG_CLR_LIMIT_BLIT_WG=1" "DEBUG_CLR_LIMIT_BLIT_WG=8" "DEBUG_CLR_LIMIT_BLIT_WG=64" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2"}; do
    echo "$v: $(env $v timeout 300 python3 tools/zstd_stream_probe.py 2>&1 | grep -E 'device_batch|all columns|rror' | cut -c1-70 | tr '\n' '|')"
  done
done
