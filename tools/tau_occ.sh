#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 2 1; do for l in 10000 13000 20000 40000 80000; do
  export DDP_HIP_TAU_MODE=$m DDP_HIP_TAU_LDS=$l
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/po -- python3 tools/lin_only.py 16 > gpurun_out/po.log 2>&1
  echo "mode $m lds $l"; python3 tools/summarize_profile.py gpurun_out/po_s.txt --stats gpurun_out/po | grep "static_tau.*true"; rm -rf gpurun_out/po
done; done
