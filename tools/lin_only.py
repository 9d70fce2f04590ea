"""development: time the linearisation kernels alone (rocprofv3 around this gives the per-kernel split)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ddp_pinocchio_amd import capi
S, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 200
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # 1: analytic jacobians + forward differences of them
model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
ctx = capi.Context(capi.ProblemSpec(model, T, batch=S, fd_mode=MODE, first_order_fd=1 if MODE == 2 else 0))
us = 0.1 * np.random.default_rng(0).normal(size=(S, T * 38))
ctx.upload("X", np.zeros((S, (T + 1) * 76))); ctx.upload("U", us); ctx.rollout()
ctx.linearize()
t0 = time.perf_counter(); ctx.linearize(); print("linearize ms", (time.perf_counter() - t0) * 1e3)
