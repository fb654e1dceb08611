#!/bin/bash
# all columns of a BGZF FASTQ as DataChunks: the host mirror (round 5) against the copy behind the scan, inside ONE box
#   gpurun -- 'bash tools/gz_chunks_ab.sh'            (GZC_GB: compressed GB, default 4)
for rep in 1 2; do
  echo "== host mirror"; python3 tools/gz_chunks_probe.py 2>&1 | grep -v amdgpu.ids
  echo "== EXG_NO_HOST_MIRROR=1"; EXG_NO_HOST_MIRROR=1 python3 tools/gz_chunks_probe.py 2>&1 | grep -v amdgpu.ids
done
