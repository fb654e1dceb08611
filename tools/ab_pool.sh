#!/bin/bash
# A/B of two library builds on the small-query and the BGZF legs (inside one box): tools/ab_pool.sh a.so b.so
bash tools/ab_lib.sh "python tools/config1_probe.py 2>/dev/null | grep COUNT" "$@"
bash tools/ab_lib.sh "python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c \"import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['configs']; print({k:c[k] for k in c if 'config1' in k or 'config4' in k})\"" "$@"
