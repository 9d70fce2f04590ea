#!/bin/bash
# Round-3 counters of the reference drivers' derivative mode (bench.py --fd-mode 1) on the GPU box (through gpurun, from the repo
# root): separate --pmc passes (kernel trace only) for the HBM traffic of the backward kernels and the issue / LDS counters of the
# analytic evaluation kernel.  The summary lands in gpurun_out/prof_r03_mode1/ and is copied into profiles/ afterwards.
R=${1:-r03}
ROOT=$(pwd)
O=$ROOT/gpurun_out/prof_${R}_mode1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3m_*
A="--fd-mode 1 --no-extra --no-cpu-baseline --steps 1 --warmup 1 --no-kernel-events"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p3m_fetch -- python3 $ROOT/bench.py $A > $O/pmc_fetch.log 2>&1 || exit 1
echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p3m_write -- python3 $ROOT/bench.py $A > $O/pmc_write.log 2>&1 || exit 1
echo write done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/p3m_sq -- python3 $ROOT/bench.py $A > $O/pmc_sq.log 2>&1 || exit 1
echo sq done
cd $ROOT
python3 tools/summarize_profile.py $O/summary_pmc_${R}_mode1.txt --pmc /tmp/p3m_fetch --pmc /tmp/p3m_write --pmc /tmp/p3m_sq --filter "bwd_,ana_eval,lin_static,forward_kernel" --note "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py $A: three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), mean per dispatch over the 2 iterations of each pass; 64 seeds, analytic first order + fd_mode 1"
grep -A3 "bwd_contract_half\|ana_eval" $O/summary_pmc_${R}_mode1.txt | head -40 | cut -c1-220
