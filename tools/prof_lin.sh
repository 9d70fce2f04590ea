#!/bin/bash
# development: per-kernel durations of one linearisation (tools/lin_only.py <seeds> <fd_mode>) under rocprofv3
# usage (on the GPU box, from the repo root): tools/prof_lin.sh <seeds> <fd_mode> <tag>
set -e
ROOT=$(pwd)
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl_$3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$3 -- python3 $ROOT/tools/lin_only.py $1 $2 > $ROOT/gpurun_out/ab/$3.log 2>&1
f=$(ls /tmp/pl_$3/*/*_kernel_stats.csv)
grep 'linearize ms' $ROOT/gpurun_out/ab/$3.log
python3 - "$f" <<'PY' | tee $ROOT/gpurun_out/ab/$3_kernels.txt
import csv, re, sys
for r in list(csv.reader(open(sys.argv[1])))[1:]:
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", r[0]); name = re.sub(r"\((LinParams|AnaParams).*$", "", name)
    if float(r[2]) > 2e5:
        print(f"  {name:62s} calls {int(r[1]):4d}  avg {float(r[3])/1e3:10.1f} us  total {int(r[2])/1e6:9.2f} ms")
PY
