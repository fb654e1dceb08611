#!/bin/bash
# zlib parity of the inflate paths, then the three token mixes A/B over library builds: tools/ab_inflate3.sh a.so b.so ...
timeout 900 python -m pytest tests/test_inflate_gpu.py tests/test_inflate_stream_gpu.py -x -q 2>&1 | tail -3
for d in fastq vcf fasta; do
  echo "== $d"
  INFLATE_DATA=$d bash tools/ab_lib.sh "INFLATE_K=32 python tools/bench_inflate.py" "$@" 2>&1 | sed 's/"members.*"ms"/"ms"/' 
done
