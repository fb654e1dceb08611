timeout 900 python -m pytest tests/test_inflate_stream_gpu.py tests/test_table_function_gpu.py -x -q 2>&1 | tail -3
GZ_RECORDS=3200000 EXG_TRACE=1 GZ_ONLY_SINGLE=1 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|inflate stream" | tail -9
