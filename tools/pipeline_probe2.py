"""development: two half-batch contexts in explicit counter-phase -- while one linearises, the other runs its backward + forward
sweeps; the host threads meet at a barrier after every phase.  usage: pipeline_probe2.py <seeds> <fd_mode> [iters]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ddp_pinocchio_amd import capi
S, MODE = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
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

    def lin(self):
        self.ctx.linearize()

    def sweeps(self):
        c = self.ctx
        rc, self.reg, self.mu, _ = c.backward(self.reg, self.mu)
        rc, step, dcost = c.forward(self.mu, n_alpha=8)
        self.reg = np.where(step >= 0.5, np.where(self.reg / 2 < 1e-5, 0.0, self.reg / 2), self.reg)
        c.swap_traj()


per = S // 2
its = [It(make(list(range(g * per, (g + 1) * per))), per) for g in range(2)]
for it in its:
    it.lin(); it.sweeps(); it.ctx.synchronize()
bar = threading.Barrier(2)
times = {"lin": [0.0, 0.0], "sweeps": [0.0, 0.0]}


def run(g):
    it = its[g]
    if g == 0:
        it.lin()                     # A starts one phase ahead
    bar.wait()
    for k in range(K):
        t0 = time.perf_counter()
        if g == 0:
            it.sweeps(); times["sweeps"][0] += time.perf_counter() - t0
        else:
            it.lin(); times["lin"][1] += time.perf_counter() - t0
        bar.wait()
        t0 = time.perf_counter()
        if g == 0:
            it.lin(); times["lin"][0] += time.perf_counter() - t0
        else:
            it.sweeps(); times["sweeps"][1] += time.perf_counter() - t0
        bar.wait()
    it.ctx.synchronize()


t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(g,)) for g in range(2)]
for t in ths:
    t.start()
for t in ths:
    t.join()
el = time.perf_counter() - t0
print(f"seeds {S} mode {MODE} counter-phase: {S * K / el:.1f} iterations/s, {el / K * 1e3:.1f} ms per iteration of all seeds; "
      f"per phase: lin {[round(v / K * 1e3, 1) for v in times['lin']]} sweeps {[round(v / K * 1e3, 1) for v in times['sweeps']]} ms")
