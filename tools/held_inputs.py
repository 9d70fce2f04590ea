"""Development helper (tools/probe_constrained.py): posture-holding trajectories generated on the device through the library's own
entry points.  Kept for the record of what was tried for the constrained benchmark lines: on these inputs the
open-loop linearisation is so unstable (light distal links) that NO sweep survives T = 200 in double, tensor-free included;
at T = 60 the config-constrained full-DDP sweep runs with 0-1 restarts.

A free rollout of the Talos-like tree under u ~ N(0, 0.1^2) is a free fall (|q| reaches 20 rad within the horizon): a
constraint to a fixed target then sits so far off that sum_i (mu eq_i) eq_xx(i) outweighs mu eq_x^T eq_x and Q_uu never turns
positive definite -- on the CPU restatement of the reference's algorithm as on the device.  The constrained
lines therefore start from a robot HOLDING a posture under computed-torque feedback and actuation noise:
    u_t = u_hold + w_t + K (x_t - x_hold),  K = -M(q0) [kp I | kd I],  w_t ~ N(0, sigma^2),  u_hold = RNEA(q0, 0, 0)
which is exactly what forward_pass computes from a constant reference trajectory, feed-forward terms w_t and gains K with the
line search off (ddp_fwd.ipp:39-51,61-63): ddp_hip_forward(n_alpha = 0) on an unconstrained context.  The gains are stiff
(kp = 2500, kd = 100: the discrete closed loop [[1, dt], [-dt kp, 1 - dt kd]] has |lambda| = 0.5 at dt = 0.01) because the
feedback is linearised at q0 only: with the usual kp = 100 the gravity gradient of the heavy base outweighs kp M_ii of the
light distal links and the loop diverges (measured on the CPU restatement: 1e147 after 200 steps).  M(q0) and u_hold come
from the point evaluations of the Model concept (ddp_hip_model_aba / _aba_derivatives)."""
import numpy as np

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddp_pinocchio_amd import capi


def held_trajectories(model, T, seeds, device=0, sigma=0.5, kp=2500.0, kd=100.0, q0=None, dt=0.01):
    """-> (xs [S][(T+1) nx], us [S][T m], q0): closed-loop rollouts of the posture-holding robot, one per global seed"""
    m = model.nv
    nx = 2 * m
    S = len(seeds)
    q0 = np.zeros(m) if q0 is None else np.asarray(q0, dtype=np.float64)
    with capi.ModelHandle(model, device) as h:
        zero = np.zeros(m)
        a0 = h.aba(q0, zero, zero)
        Minv = h.aba_derivatives(q0, zero, zero)[2]
    M = np.linalg.inv(0.5 * (Minv + Minv.T))
    u_hold = -M @ a0                                     # aba(q0, 0, u) = a0 + M^-1 u = 0
    K = -np.hstack([kp * M, kd * M])                     # m x n
    spec = capi.ProblemSpec(model, T, dt=dt, c=1.0, batch=S, fd_mode=0, first_order_fd=1, eq_kind=capi.EQ_NONE,
                            ne=np.zeros(T, dtype=np.int64))
    x_hold = np.concatenate([q0, np.zeros(m)])
    with capi.Context(spec, device=device, flags=capi.FLAG_NO_TENSORS) as ctx:
        X = np.tile(x_hold, (S, T + 1))
        U = np.tile(u_hold, (S, T))
        ctx.upload("X", X); ctx.upload("U", U); ctx.upload("X_NEW", X); ctx.upload("U_NEW", U)
        ctx.upload("FB_ORIGIN", np.tile(x_hold, (S, T)))
        ctx.upload("FB_VAL", np.stack([sigma * np.random.default_rng(0xDD9000 + 5000 + g).normal(size=T * m) for g in seeds]))
        ctx.upload("FB_JAC", np.tile(K.T.reshape(-1), (S, T)))      # K_t column-major m x n
        ctx.forward(np.ones(S), n_alpha=0)
        xs, us = ctx.download("X_NEW"), ctx.download("U_NEW")
    return xs, us, q0
