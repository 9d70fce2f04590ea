#!/bin/bash
# development: configuration-level workspace slice size (instances x t per launch pair)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for q in 64 128 256 512 1024 4096; do
  export DDP_HIP_QWS_BT=$q
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pw -- python3 tools/lin_only.py 64 > gpurun_out/pw.log 2>&1
  python3 tools/summarize_profile.py gpurun_out/pw_s.txt --stats gpurun_out/pw > /dev/null
  echo "slice $q: $(grep -h 'cfg_up\|cfg_down' gpurun_out/pw_s.txt | awk '{s+=$(NF-4)} END {print s/2}') ms per linearisation (cfg up+down); $(grep 'linearize ms' gpurun_out/pw.log)"; rm -rf gpurun_out/pw
done
