# single-member gzip: chunk size A/B inside one gpurun call
timeout 900 python -m pytest tests/test_inflate_stream_gpu.py -x -q 2>&1 | tail -3
for c in 0 196608 131072; do
  echo "chunk $c"
  if [ $c = 0 ]; then unset EXG_STREAM_CHUNK_BYTES; else export EXG_STREAM_CHUNK_BYTES=$c; fi
  GZ_RECORDS=3200000 EXG_TRACE=1 GZ_ONLY_SINGLE=1 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|inflate stream" | tail -9
done
