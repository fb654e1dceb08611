GZ_RECORDS=3200000 EXG_TRACE=1 GZ_ONLY_SINGLE=1 timeout 300 python tools/gz_probe.py 2>&1 | grep -E "single member|searched|decoded|in all" | tail -4
