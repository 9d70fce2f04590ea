#!/bin/bash
# development: small-batch lines (config 3 = 1 seed, config 4's per-GPU load at N = 8 = 8 seeds) with / without K5 beside K3
mkdir -p gpurun_out/r03b
for S in 1 8 16; do
  for F in 0 1; do
    if [ $F -eq 1 ]; then export DDP_HIP_BWD_FORK=1; else unset DDP_HIP_BWD_FORK; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra --seeds-per-gpu $S --steps 5 > gpurun_out/r03b/bench_s${S}_f$F.json 2>/dev/null
    python3 - $S $F <<'PY'
import json, sys
s, f = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r03b/bench_s{s}_f{f}.json").read().strip().splitlines()[-1])
print("seeds", s, "fork", f, round(d["value"], 1), {k: round(v, 2) for k, v in d["phases_ms_per_step"].items()}, "plain", d["uninstrumented"] and {k: round(v, 2) for k, v in d["uninstrumented"]["phases_ms_per_step"].items()})
PY
  done
done
