#!/bin/bash
# development run on the GPU box: GPU test suite, then the default bench line (with the extra legs), no CPU baseline
O=gpurun_out/r03a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1
rc=$?
tail -15 $O/gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
rc=$?
echo "bench rc $rc"; tail -3 $O/bench_default.err
python3 - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r03a/bench_default.json').read().strip().splitlines()[-1])
    print(d['value'], d['phases_ms_per_step'], d.get('best_pick'))
    for k, v in d.get('extra', {}).items():
        print(k, json.dumps(v)[:900])
except Exception as e:
    print('no bench line', e)
PY
