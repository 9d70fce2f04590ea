#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun); summaries land in gpurun_out/ and are then
# copied into profiles/ by hand.
set -e
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --seeds-per-gpu ${S:-16} --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bwd64_stats -- python3 tools/dev_bwd_timing.py --batch 64 --reps 2 > $O/bwd64.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bwd64_pmc_fetch -- python3 tools/dev_bwd_timing.py --batch 64 --reps 1 > $O/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bwd64_pmc_write -- python3 tools/dev_bwd_timing.py --batch 64 --reps 1 > $O/pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/bwd64_pmc_sq -- python3 tools/dev_bwd_timing.py --batch 64 --reps 1 > $O/pmc3.log 2>&1
python3 tools/summarize_profile.py $O/summary_bench_$R.txt --stats $O/bench_stats --note "bench.py --seeds-per-gpu ${S:-16} --steps 2 --warmup 1 --no-cpu-baseline (full DDP iteration, Talos-like, T=200): $(grep '^{' $O/bench.log | tail -1 | head -c 1500)"
python3 tools/summarize_profile.py $O/summary_bwd64_$R.txt --stats $O/bwd64_stats --pmc $O/bwd64_pmc_fetch --pmc $O/bwd64_pmc_write --pmc $O/bwd64_pmc_sq --note "tools/dev_bwd_timing.py --batch 64 (backward sweep alone, random derivative inputs; 'assemble' = K3 bwd_contract, 'gains' = K4 bwd_riccati; HIP-event timing): $(grep -h '^batch 64\|^assemble\|^gains' $O/bwd64.log | tr '\n' ' ')"
