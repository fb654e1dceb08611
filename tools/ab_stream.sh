for pin in 0 1; do
  if [ $pin = 1 ]; then unset EXG_NO_NUMA_PIN; else export EXG_NO_NUMA_PIN=1; fi
  echo "pin=$pin"; GZ_RECORDS=3200000 EXG_TRACE=1 GZ_ONLY_SINGLE=1 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|gz:|in all|searched|decoded" | tail -8
done
