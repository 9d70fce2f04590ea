#!/bin/bash
# PMC passes over the linearisation kernels alone (tools/lin_only.py); run through gpurun.
set -e
S=${S:-16}
TAG=${1:-lin}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/lin_only.py $S > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_sq -- python3 tools/lin_only.py $S > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq2 -- python3 tools/lin_only.py $S > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/lin_only.py $S > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/lin_only.py $S > $O/pmc_w.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python3 tools/lin_only.py $S > $O/pmc_t.log 2>&1
python3 tools/summarize_profile.py $O/summary_$TAG.txt --stats $O/stats --pmc $O/pmc_sq --pmc $O/pmc_sq2 --pmc $O/pmc_fetch --pmc $O/pmc_write --pmc $O/pmc_tcc --note "tools/lin_only.py $S: $(grep -h 'linearize ms' $O/stats.log)"
cat $O/summary_$TAG.txt
