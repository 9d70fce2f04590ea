"""development: phase times of one DDP iteration on BASELINE config 2 (UR5-like chain, T = 100, config constraint every step)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddp_pinocchio_amd import capi
from problems import make

for name, T, B, fd_mode in (("chain6", 100, 1, 0), ("chain6", 100, 64, 0), ("chain6", 10, 1, 2), ("pendulum", 50, 1, 2)):
    model, spec, o = make(name, T, batch=B, fd_mode=fd_mode)
    with capi.Context(spec, flags=0 if fd_mode else capi.FLAG_NO_TENSORS) as ctx:
        nx, m = o.nx, o.m
        us = 0.01 * np.random.default_rng(1).normal(size=(B, T * m))
        ctx.upload("X", np.zeros((B, (T + 1) * nx))); ctx.upload("U", us); ctx.rollout()
        xs = ctx.download("X")
        ctx.upload("X_NEW", xs); ctx.upload("U_NEW", us)
        if o.Etot:
            ctx.upload("MULT_ORIGIN", np.ascontiguousarray(xs[:, :T * nx])); ctx.upload("MULT_VAL", np.zeros((B, o.Etot)))
            ctx.upload("MULT_JAC", 0.01 * np.random.default_rng(2).normal(size=(B, o.Etot * o.n)))
        mu = np.full(B, 1e3); reg = np.zeros(B)
        ph = np.zeros(5)
        for it in range(4):
            t0 = time.perf_counter(); ctx.linearize(); t1 = time.perf_counter()
            ctx.update_origin(0); ctx.update_origin(1); ctx.optimality(mu); t2 = time.perf_counter()
            rc, reg, mu, rs = ctx.backward(reg, mu, 30); t3 = time.perf_counter()
            rc, step, dc = ctx.forward(mu, n_alpha=8); t4 = time.perf_counter()
            ctx.swap_traj()
            if it: ph += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
        ph *= 1e3 / 3
        print(f"{name} T={T} B={B} fd_mode={fd_mode}: linearise {ph[0]:.2f}  outer {ph[1]:.2f}  backward {ph[2]:.2f}  forward {ph[3]:.2f}  total {ph[4]:.2f} ms; paths {ctx.info()}", flush=True)
