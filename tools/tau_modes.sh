#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ${MODES:-0 1 2}; do
  export DDP_HIP_TAU_MODE=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pm$m -- python3 tools/lin_only.py 16 > gpurun_out/pm$m.log 2>&1
  echo "mode $m"; python3 tools/summarize_profile.py gpurun_out/pm${m}_s.txt --stats gpurun_out/pm$m | grep static
done
