"""development probe: the constrained Talos-size problems at T = 200, tensor-free and full, on held-posture trajectories"""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddp_pinocchio_amd import capi
import held_inputs as synth_inputs

S = 4
model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
m = model.nv; nx = 2 * m; n = nx
for T in (200, 60):
  xs, us, q0 = synth_inputs.held_trajectories(model, T, list(range(S)))
  for name in ("tree38_config", "tree38_frame"):
    for fd_mode in (0, 2):
        for jac_sigma, mu0 in ((0.01, 1e3), (0.01, 1e1), (0.0, 1e3)):
            if name == "tree38_config":
                ne = np.full(T, 38, dtype=np.int64)
                kw = dict(eq_kind=capi.EQ_CONFIG, eq_advance=2, ne=ne, eq_target=np.tile(q0, T))
            else:
                ne = np.zeros(T, dtype=np.int64); ne[T - 2] = 3
                with capi.ModelHandle(model) as h:
                    p3, _ = h.frame(27, np.array([0.0, 0.0, 0.1]), q0)
                kw = dict(eq_kind=capi.EQ_FRAME, eq_advance=2, ne=ne, eq_target=p3 + np.array([0.05, -0.02, 0.03]), frame_joint=27, frame_off=(0.0, 0.0, 0.1))
            spec = capi.ProblemSpec(model, T, dt=0.01, c=1.0, batch=S, fd_mode=fd_mode, first_order_fd=1, **kw)
            Etot = int(ne.sum())
            log = []
            try:
                with capi.Context(spec, flags=0 if fd_mode else capi.FLAG_NO_TENSORS) as ctx:
                    ctx.upload("X", xs); ctx.upload("U", us); ctx.upload("X_NEW", xs); ctx.upload("U_NEW", us)
                    ctx.upload("MULT_ORIGIN", np.ascontiguousarray(xs[:, :T * nx]))
                    ctx.upload("MULT_VAL", np.zeros((S, Etot)))
                    ctx.upload("MULT_JAC", np.stack([jac_sigma * np.random.default_rng(6000 + g).normal(size=Etot * n) for g in range(S)]))
                    mu = np.full(S, mu0); reg = np.zeros(S)
                    for it in range(3):
                        t0 = time.perf_counter(); ctx.linearize(); t1 = time.perf_counter()
                        if it: ctx.update_origin(0); ctx.update_origin(1)
                        rc, reg, mu, restarts = ctx.backward(reg, mu, 100)
                        t2 = time.perf_counter()
                        rc2, step, dc = ctx.forward(mu, n_alpha=8)
                        t3 = time.perf_counter()
                        ctx.swap_traj()
                        log.append((list(map(int, restarts)), [float(s) for s in step], round((t1-t0)*1e3,1), round((t2-t1)*1e3,1), round((t3-t2)*1e3,1)))
                    print(T, name, fd_mode, jac_sigma, mu0, "OK", log, flush=True)
            except Exception as e:
                print(T, name, fd_mode, jac_sigma, mu0, "FAIL", repr(e), log, flush=True)
