#!/bin/bash
# tuning sweep of the K3 job sizes (development): x-column groups of CBX slabs, u-column groups of CBU slabs
for cfg in ${CFGS:-"3 9" "4 8" "4 9" "4 10" "3 8" "2 6" "4 12" "5 10" "6 13" "2 13"}; do
  set -- $cfg
  echo "cbx=$1 cbu=$2: $(DDP_HIP_BWD_CBX=$1 DDP_HIP_BWD_CBU=$2 timeout -k 10 300 python tools/dev_bwd_timing.py --batch ${B:-64} --reps 2 2>&1 | grep assemble)"
done
