#!/bin/bash
# development: per-kernel durations of the linearisation for A/B builds of the library (build_ab/libddp_hip_<tag>.so)
# usage (on the GPU box, from the repo root): tools/ab_kernels.sh <seeds> <tag> [<tag> ...]     tag "base" = the in-tree library
set -e
ROOT=$(pwd)
SEEDS=$1; shift
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = base ]; then unset DDP_HIP_LIB; else export DDP_HIP_LIB=$ROOT/build_ab/libddp_hip_$tag.so; fi
  rm -rf /tmp/ab_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$tag -- python3 $ROOT/tools/lin_only.py $SEEDS > $ROOT/gpurun_out/ab/$tag.log 2>&1
  f=$(ls /tmp/ab_$tag/*/*_kernel_stats.csv)
  echo "== $tag: $(grep 'linearize ms' $ROOT/gpurun_out/ab/$tag.log)" | tee -a $ROOT/gpurun_out/ab/summary.txt
  python3 - "$f" <<'PY' | tee -a $ROOT/gpurun_out/ab/summary.txt
import csv, re, sys
for r in list(csv.reader(open(sys.argv[1])))[1:]:
    if "lin_" in r[0]:
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r[0]); name = re.sub(r"\((LinParams).*$", "", name)
        print(f"  {name:58s} calls {int(r[1]):4d}  avg {float(r[3])/1e3:10.1f} us  total {int(r[2])/1e6:9.2f} ms")
PY
done
