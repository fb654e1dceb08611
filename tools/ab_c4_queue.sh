#!/bin/bash
# config 4 under source-queue depths x lanes, inside ONE box:   gpurun -- 'bash tools/ab_c4_queue.sh'
export C4_GB=${C4_GB:-10} GPU_MAX_HW_QUEUES=8
timeout 300 python3 tools/c4_probe.py build || exit 1
for l in ${C4_LANES:-3 4}; do for q in ${C4_SRCQ:-2 3 4 6}; do
  echo -n "lanes=$l source-queue=$q: "; EXG_GZ_LANES=$l EXG_SOURCE_QUEUE=$q timeout 120 python3 tools/c4_probe.py run 2>&1 | grep COUNT
done; done
rm -f /dev/shm/exg_c4.fastq.gz
