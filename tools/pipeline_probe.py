"""development: do two half-batches on contexts (streams) of their own, driven by two host threads, overlap?  One half's
linearisation (throughput-bound: fills the GPU) should hide the other half's backward + forward sweeps (latency-bound: a few
hundred waves).  usage: pipeline_probe.py <seeds> <groups> <fd_mode> [iters]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ddp_pinocchio_amd import capi
S, G, MODE = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4
T, nv = 200, 38
model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)


def make(seeds):
    spec = capi.ProblemSpec(model, T, dt=0.01, c=1.0, batch=len(seeds), fd_mode=MODE, first_order_fd=0 if MODE == 1 else 1)
    ctx = capi.Context(spec)
    us = np.stack([0.1 * np.random.default_rng(0xDD9000 + 3000 + g).normal(size=T * nv) for g in seeds])
    ctx.upload("X", np.zeros((len(seeds), (T + 1) * 2 * nv))); ctx.upload("U", us); ctx.rollout()
    ctx.upload("X_NEW", ctx.download("X")); ctx.upload("U_NEW", us)
    return ctx


class It:
    def __init__(self, ctx, n):
        self.ctx, self.reg, self.mu = ctx, np.zeros(n), np.full(n, 1e2)

    def step(self):
        c = self.ctx
        c.linearize()
        rc, self.reg, self.mu, _ = c.backward(self.reg, self.mu)
        rc, step, dcost = c.forward(self.mu, n_alpha=8)
        self.reg = np.where(step >= 0.5, np.where(self.reg / 2 < 1e-5, 0.0, self.reg / 2), self.reg)
        c.swap_traj()


per = S // G
its = [It(make(list(range(g * per, (g + 1) * per))), per) for g in range(G)]
for it in its:
    it.step()
for it in its:
    it.ctx.synchronize()


STAGGER = float(sys.argv[5]) * 1e-3 if len(sys.argv) > 5 else 0.0


def run(it, k, delay=0.0):
    if delay:
        time.sleep(delay)
    for _ in range(k):
        it.step()
    it.ctx.synchronize()


t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(it, K, g * STAGGER)) for g, it in enumerate(its)]
for t in ths:
    t.start()
for t in ths:
    t.join()
el = time.perf_counter() - t0
print(f"seeds {S} groups {G} mode {MODE}: {S * K / el:.1f} iterations/s, {el / K * 1e3:.1f} ms per iteration of all seeds")
