"""development: one line out of a bench.py JSON file"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["phases_ms_per_step"], (d.get("uninstrumented") or {}).get("iterations_per_s_this_rank"))
print(d.get("kernels_ms_per_step"))
