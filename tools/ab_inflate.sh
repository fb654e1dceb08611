# A/B of the inflate window policies inside one gpurun call (box-to-box spread is ~10 %)
timeout 900 python -m pytest tests/test_inflate_gpu.py tests/test_inflate_stream_gpu.py tests/test_table_function_gpu.py -x -q 2>&1 | tail -3
for r in 2048 4096 32768; do echo "BGZF ring $r"; EXG_INFLATE_RING=$r INFLATE_K=64 timeout 200 python tools/bench_inflate.py 2>&1 | tail -1; done
for r in 2048 4096 32768; do echo "stream ring $r"; EXG_STREAM_RING=$r GZ_RECORDS=3200000 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|bgzf" | tail -2; done
