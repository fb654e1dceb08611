GZ_SOAK_GB=4 timeout 600 python tools/gz_soak.py 2>&1 | tail -5
GZ_SOAK_GB=10 GZ_SOAK_CHUNKS=0 timeout 600 python tools/gz_soak.py 2>&1 | tail -3
GZ_RECORDS=3200000 EXG_TRACE=1 GZ_ONLY_SINGLE=1 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|inflate stream" | tail -8
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/inf_pmc3 -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_inflate.py > $GRAFT_REPO_ROOT/gpurun_out/inf_pmc3.log 2>&1
