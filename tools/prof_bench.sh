#!/bin/bash
# development: rocprofv3 kernel stats of a bench.py run; usage (GPU box, repo root): tools/prof_bench.sh <tag> [bench args...]
set -e
ROOT=$(pwd); TAG=$1; shift
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $ROOT/bench.py --no-extra --cpu-iterations 0 "$@" > $ROOT/gpurun_out/prof/$TAG.log 2>&1
cp /tmp/prof_$TAG/*/*_kernel_stats.csv $ROOT/gpurun_out/prof/${TAG}_kernel_stats.csv
tail -1 $ROOT/gpurun_out/prof/$TAG.log | cut -c1-400
python3 $ROOT/tools/summarize_profile.py $ROOT/gpurun_out/prof/${TAG}_summary.txt --stats /tmp/prof_$TAG --filter "" && head -32 $ROOT/gpurun_out/prof/${TAG}_summary.txt | cut -c1-130
