#!/bin/bash
# config 4's all-columns drain with the reader's timeline (EXG_TRACE=2) -> gpurun_out/c4_trace.txt
#   gpurun -- 'bash tools/c4_trace.sh'      C4_GB=10 by default
export C4_GB=${C4_GB:-10}
mkdir -p gpurun_out
timeout 300 python3 tools/c4_probe.py build || exit 1
EXG_TRACE=2 timeout 200 python3 tools/c4_probe.py run > gpurun_out/c4_trace.txt 2>&1
tail -2 gpurun_out/c4_trace.txt
rm -f /dev/shm/exg_c4.fastq.gz
