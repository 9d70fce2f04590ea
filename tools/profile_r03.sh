#!/bin/bash
# Round-3 profiles on the GPU box (through gpurun, from the repo root): kernel stats of the bench command, then separate --pmc
# passes (kernel trace only) for the HBM traffic of the backward kernels and the issue / LDS counters of the stencil kernels.
# Summaries land in gpurun_out/prof_r03/ and are copied into profiles/ afterwards.
R=${1:-r03}
ROOT=$(pwd)
O=$ROOT/gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3_*
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3_stats -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline > $O/bench_stats.log 2>&1 || exit 1
echo stats done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p3_fetch -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline --steps 1 --warmup 1 --no-kernel-events > $O/pmc_fetch.log 2>&1 || exit 1
echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p3_write -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline --steps 1 --warmup 1 --no-kernel-events > $O/pmc_write.log 2>&1 || exit 1
echo write done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/p3_sq -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline --steps 1 --warmup 1 --no-kernel-events > $O/pmc_sq.log 2>&1 || exit 1
echo sq done
cd $ROOT
python3 tools/summarize_profile.py $O/summary_bench_$R.txt --stats /tmp/p3_stats --filter "" --note "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-extra --no-cpu-baseline (1 warm-up + 3 timed iterations, 64 seeds, full DDP mode 2); the bench line of this run: $(grep '^{' $O/bench_stats.log | tail -1 | head -c 2600)"
python3 tools/summarize_profile.py $O/summary_pmc_$R.txt --pmc /tmp/p3_fetch --pmc /tmp/p3_write --pmc /tmp/p3_sq --filter "bwd_,lin_static,forward_kernel" --note "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --no-extra --no-cpu-baseline --steps 1 --warmup 1 --no-kernel-events: three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), mean per dispatch over the 2 iterations of each pass; 64 seeds, full DDP mode 2"
head -40 $O/summary_bench_$R.txt | cut -c1-150
grep -A3 "bwd_contract_half" $O/summary_pmc_$R.txt | head -30 | cut -c1-200
