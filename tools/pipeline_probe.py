"""development probe: P contexts (S/P seeds each) driven from P host threads, staggered by half an iteration"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ddp_pinocchio_amd import capi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T = 200
model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
ctxs = []
for k in range(P):
    s = S // P
    ctx = capi.Context(capi.ProblemSpec(model, T, batch=s, fd_mode=2))
    us = 0.1 * np.random.default_rng(k).normal(size=(s, T * 38))
    ctx.upload("X", np.zeros((s, (T + 1) * 76))); ctx.upload("U", us); ctx.rollout()
    ctx.upload("X_NEW", ctx.download("X")); ctx.upload("U_NEW", us)
    ctxs.append(ctx)
def run(k, steps, start_evt, lin_done):
    ctx = ctxs[k]; s = S // P
    reg = np.zeros(s); mu = np.full(s, 1e2)
    start_evt.wait()
    for it in range(steps):
        ctx.linearize()
        lin_done.set()
        rc, reg, mu, r = ctx.backward(reg, mu)
        rc, step, dc = ctx.forward(mu, n_alpha=8)
        ctx.swap_traj()
def go(steps):
    evts = [threading.Event() for _ in range(P)]
    lins = [threading.Event() for _ in range(P)]
    th = [threading.Thread(target=run, args=(k, steps, evts[k], lins[k])) for k in range(P)]
    for t in th: t.start()
    t0 = time.perf_counter()
    evts[0].set()
    for k in range(1, P):
        lins[k - 1].wait()      # stagger: context k starts when context k-1 has finished its first linearisation
        evts[k].set()
    for t in th: t.join()
    return time.perf_counter() - t0
go(1)
el = go(STEPS)
print(f"S={S} P={P} steps={STEPS}: {el / STEPS * 1e3:.1f} ms/step  {S * STEPS / el:.1f} it/s")
