# A/B of the inflate variants inside one gpurun call (box-to-box spread is ~10 %); needs a development build of the library:
# EXG_CXXFLAGS=-DEXG_DEV_PROBE python __graft_entry__.py build
timeout 900 python -m pytest tests/test_inflate_gpu.py tests/test_inflate_stream_gpu.py tests/test_table_function_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
for e in 0 1 2 4; do echo "BGZF emit $e"; EXG_INFLATE_EMIT=$e INFLATE_K=64 timeout 200 python tools/bench_inflate.py 2>&1 | tail -1; done
