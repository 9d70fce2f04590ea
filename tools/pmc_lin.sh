#!/bin/bash
# development: LDS / wait counters of the linearisation kernels (separate --pmc pass, kernel trace only)
set -e
ROOT=$(pwd); S=${1:-16}
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lin
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_lin -- python3 $ROOT/tools/lin_only.py $S > $ROOT/gpurun_out/pmc/lin.log 2>&1
cd $ROOT
python3 tools/summarize_profile.py gpurun_out/pmc/summary_lin_pmc.txt --pmc /tmp/pmc_lin --filter "lin_"
grep -E "BANK_CONFLICT|IDX_ACTIVE|WAVE_CYCLES|INST_VALU" gpurun_out/pmc/summary_lin_pmc.txt | cut -c1-140
