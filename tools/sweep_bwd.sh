#!/bin/bash
# tuning sweep of the assemble job sizes (development)
for cfg in "3 9" "2 6" "4 12" "2 13" "3 13" "4 9" "6 16" "1 4"; do
  set -- $cfg
  echo "cbx=$1 cbu=$2: $(DDP_HIP_BWD_CBX=$1 DDP_HIP_BWD_CBU=$2 timeout -k 10 300 python tools/dev_bwd_timing.py --batch ${B:-64} --reps 2 2>&1 | grep assemble)"
done
