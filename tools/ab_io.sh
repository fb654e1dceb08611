# NUMA pinning of the reader's I/O threads, A/B in separate processes
for rep in 1 2; do
echo "pin:";    timeout 300 python tools/reader_probe.py 2>&1 | grep -E "^count|^arrow|^chunks" | awk 'NR%2==0' | cut -d' ' -f1-2,4-6 | tr '\n' ';'; echo
echo "no pin:"; EXG_NO_NUMA_PIN=1 timeout 300 python tools/reader_probe.py 2>&1 | grep -E "^count|^arrow|^chunks" | awk 'NR%2==0' | cut -d' ' -f1-2,4-6 | tr '\n' ';'; echo
done
