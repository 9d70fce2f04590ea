#!/bin/bash
# end-of-round runs on the GPU box (through gpurun, from the repo root): the GPU test suite, the default bench line, the bench
# variants quoted in DESIGN.md, and the rocprofv3 kernel summary of the bench command; outputs under gpurun_out/final/
set -e
R=${1:-r03}
O=gpurun_out/final
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1 || true
tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 python bench.py --fd-mode 1 --no-cpu-baseline --no-extra > $O/bench_fd1.json 2> $O/bench_fd1.err
timeout -k 10 300 python bench.py --mode gn --no-cpu-baseline --no-extra > $O/bench_gn.json 2> $O/bench_gn.err
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 $ROOT/bench.py --no-extra --no-cpu-baseline > $ROOT/$O/bench_rocprof.log 2>&1
cd $ROOT
python3 tools/summarize_profile.py $O/summary_bench_$R.txt --stats /tmp/prof_final --filter "" --note "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-extra --no-cpu-baseline (1 warm-up + 3 timed iterations, 64 seeds, full DDP mode 2); the bench line of this run: $(grep '^{' $O/bench_rocprof.log | tail -1 | head -c 1800)"
head -12 $O/summary_bench_$R.txt | cut -c1-160
# the same for the reference drivers' derivative mode (analytic first order + fd_mode 1)
cd /tmp
rm -rf /tmp/prof_final_m1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final_m1 -- python3 $ROOT/bench.py --fd-mode 1 --no-extra --no-cpu-baseline > $ROOT/$O/bench_rocprof_m1.log 2>&1
cd $ROOT
python3 tools/summarize_profile.py $O/summary_bench_${R}_mode1.txt --stats /tmp/prof_final_m1 --filter "" --note "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --fd-mode 1 --no-extra --no-cpu-baseline (1 warm-up + 3 timed iterations, 64 seeds, analytic first order + fd_mode 1); the bench line of this run: $(grep '^{' $O/bench_rocprof_m1.log | tail -1 | head -c 1800)"
head -10 $O/summary_bench_${R}_mode1.txt | cut -c1-160
# the N = 2 code path on this one GPU (gloo; RCCL cannot place two ranks on one device): bench.py starts its ranks itself
DDP_BENCH_BACKEND=gloo DDP_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --seeds-per-gpu 16 --no-cpu-baseline --no-extra > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank_rehearsal.err || true
