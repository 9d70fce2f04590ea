"""development: phase split of the forward latency kernel (build_ab/libddp_hip_fstamps.so, built with -DFWD_STAMPS):
accumulated s_memrealtime (100 MHz) of workgroup 0 over the 200 steps of one forward pass"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DDP_HIP_LIB"] = os.path.join(ROOT, "build_ab", "libddp_hip_fstamps.so")
sys.path.insert(0, ROOT)
import numpy as np
from ddp_pinocchio_amd import capi
S, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 200
model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
ctx = capi.Context(capi.ProblemSpec(model, T, batch=S, fd_mode=0), flags=capi.FLAG_NO_TENSORS)
us = 0.1 * np.random.default_rng(0).normal(size=(S, T * 38))
ctx.upload("X", np.zeros((S, (T + 1) * 76))); ctx.upload("U", us); ctx.rollout()
ctx.upload("X_NEW", ctx.download("X")); ctx.upload("U_NEW", us)
ctx.linearize()
ctx.backward(np.zeros(S), np.full(S, 1e2))
for _ in range(2):
    rc, step, dcost = ctx.forward(np.full(S, 1e2), n_alpha=8)
out = (C.c_ulonglong * 12)()
assert capi.lib().ddp_hip_debug_fwd_stamps(out) == 0
names = ["dx + K dx + u", "cost term", "request t+1", "placements", "pass 1", "pass 2: wait at the level barrier", "pass 3", "x update", "park",
         "pass 2: tables + inertia sums", "pass 2: U, D, 1/D", "pass 2: force half"]
tot = sum(out[:12])
for i, nme in enumerate(names):
    print(f"{nme:36s} {out[i] / 100.0 / T:8.2f} us / step  ({100.0 * out[i] / tot:5.1f} %)")
print(f"{'total':16s} {tot / 100.0 / T:8.2f} us / step")
