"""development: wall time per iteration of ddp_hip_solve on the small reference-driver shapes, with the loop enqueueing
(ddp_hip_set_async, the default) and with every call waiting (DDP_HIP_SOLVE_SYNC=1: round 2's behaviour)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddp_pinocchio_amd import capi, solver
from problems import make

for name, T, iters in (("pendulum", 50, 60), ("chain6", 10, 60), ("chain6", 100, 40), ("tree38_frame", 20, 10)):
    for mode in ("async", "sync"):
        if mode == "sync": os.environ["DDP_HIP_SOLVE_SYNC"] = "1"
        else: os.environ.pop("DDP_HIP_SOLVE_SYNC", None)
        model, spec, o = make(name, T, batch=1, fd_mode=0)
        with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
            nx, m = o.nx, o.m
            x0 = np.zeros((T + 1) * nx)
            if nx != 2 * m: x0[6] = 1.0
            us = 0.01 * np.random.default_rng(1).normal(size=T * m)
            ctx.upload("X", x0); ctx.upload("U", us); ctx.rollout()
            xs = ctx.download("X")[0]
            def load():
                ctx.upload("X", xs); ctx.upload("U", us); ctx.upload("X_NEW", xs); ctx.upload("U_NEW", us)
                if o.Etot:
                    ctx.upload("MULT_ORIGIN", xs[:T * nx]); ctx.upload("MULT_VAL", np.zeros(o.Etot))
                    ctx.upload("MULT_JAC", 0.01 * np.random.default_rng(2).normal(size=o.Etot * o.n))
            load(); solver.solve(ctx, 3, 0.0, 1e3, 0.0, 1e-1, 10.0)          # warm-up (graphs, caches)
            load(); ctx.synchronize()
            t0 = time.perf_counter()
            log = solver.solve(ctx, iters, 0.0, 1e3, 0.0, 1e-1, 10.0)      # threshold 0: never stops early
            dt = time.perf_counter() - t0
            print(f"{name:14s} T={T:4d} {mode:5s}: {dt / iters * 1e3:8.3f} ms per iteration ({int(log['iterations'][0])} iterations)", flush=True)
