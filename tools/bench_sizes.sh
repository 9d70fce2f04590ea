for s in 8 16 128; do
  timeout -k 10 300 python bench.py --seeds-per-gpu $s --no-cpu-baseline --no-extra > gpurun_out/bench_s$s.json 2>gpurun_out/bench_s$s.err
done
python - <<'PY'
import json
for s in (8,16,128):
    d=json.load(open(f"gpurun_out/bench_s{s}.json"))
    print(s, round(d["value"],1), round(d["ms_per_step"],1), {k:round(v,1) for k,v in d["phases_ms_per_step"].items()}, round(d["roofline"]["frac"],3), round(d["roofline"]["sweep_frac"],3))
PY
