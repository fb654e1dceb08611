#!/bin/bash
# config 4 under lanes / hardware-queue settings (and library builds given as arguments), inside ONE box:
#   gpurun -- 'bash tools/ab_c4.sh [a.so b.so ...]'      C4_GB=10 by default
export C4_GB=${C4_GB:-10}
timeout 300 python3 tools/c4_probe.py build || exit 1
LIBS=${@:-libexon_gpu.so}
for rep in 1 2; do
  for lib in $LIBS; do
    [ "$lib" != libexon_gpu.so ] && cp exon_duckdb_amd/lib/$lib exon_duckdb_amd/lib/libexon_gpu.so
    for q in ${C4_QUEUES:-4 8}; do for l in ${C4_LANES:-2 3 4}; do
      echo -n "$lib queues=$q lanes=$l: "; GPU_MAX_HW_QUEUES=$q EXG_GZ_LANES=$l timeout 120 python3 tools/c4_probe.py run 2>&1 | grep COUNT
    done; done
  done
done
rm -f /dev/shm/exg_c4.fastq.gz
