# end-to-end gzip numbers + inflate kernel stats (run on the GPU box from the repo root)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; TAG=${1:-r01_g}; mkdir -p $OUT; export TMPDIR=/tmp
GZ_SOAK_GB=4 timeout 600 python tools/gz_soak.py 2>&1 | tail -5
GZ_SOAK_GB=10 GZ_SOAK_CHUNKS=0 timeout 600 python tools/gz_soak.py 2>&1 | tail -3
GZ_RECORDS=3200000 EXG_TRACE=1 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|bgzf:|inflate stream" | tail -10
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_inflate -o kt --output-format csv -- python3 $ROOT/tools/bench_inflate.py > $OUT/${TAG}_kt_inflate.log 2>&1
GZ_RECORDS=800000 GZ_ONLY_SINGLE=1 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt_gzstream -o kt --output-format csv -- python3 $ROOT/tools/gz_probe.py > $OUT/${TAG}_kt_gzstream.log 2>&1
tail -1 $OUT/${TAG}_kt_inflate.log
