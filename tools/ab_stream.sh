# single-member gzip: chunk size A/B inside one gpurun call
for c in 131072 0; do
  echo "chunk $c"
  if [ $c = 0 ]; then unset EXG_STREAM_CHUNK_BYTES; else export EXG_STREAM_CHUNK_BYTES=$c; fi
  GZ_RECORDS=3200000 EXG_TRACE=1 GZ_ONLY_SINGLE=1 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|inflate stream|gz:" | tail -24
done
