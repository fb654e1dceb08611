#!/bin/bash
# end-to-end FASTQ through the reader with and without the zero-bounce upload, inside ONE box: tools/ab_e2e.sh [GB]
GB=${1:-4}
for round in 1 2 3; do
  for zb in 1 0; do
    echo "round $round EXG_ZERO_BOUNCE=$zb: $(EXG_ZERO_BOUNCE=$zb timeout 600 python tools/e2e_probe.py $GB 2>/dev/null | tail -1)"
  done
done
