#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ${MODES:-1 2}; do
  export DDP_HIP_TAU_MODE=$m
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pq$m -- python3 tools/lin_only.py 16 > gpurun_out/pq$m.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pr$m -- python3 tools/lin_only.py 16 > gpurun_out/pr$m.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/ps$m -- python3 tools/lin_only.py 16 > gpurun_out/ps$m.log 2>&1
  echo "mode $m"; python3 tools/summarize_profile.py gpurun_out/pq${m}_s.txt --pmc gpurun_out/pq$m --pmc gpurun_out/pr$m --pmc gpurun_out/ps$m | grep static
done
