"""Parity at the sizes BASELINE.json names (round-1 verdict: "configs untested"): the closed-loop forward sweep of the
Talos-like tree at T = 200, the frame-constrained chain (test/pinocchio_spatial_eq_ddp.cpp shape, config 5) at T = 200
-- linearise, backward, forward and a whole solve -- and the Talos linearisation at T = 200 against the ORACLE at the
(instance, t) pairs that straddle a workspace slice of the static-topology stencil (lin_static.hip)."""
import numpy as np
import pytest

from problems import held_trajectory, initial_trajectory, make
from synth import rel_err, stepwise_backward_check
from test_dynamics_parity import DERIV_SEQS, TENSOR_SEQS, _one_iteration_inputs, _upload_traj

EPS, E1, E2 = 2.220446049250313e-16, 1.4901161193847656e-08, 1.220703125e-04


def _upload_affine(ctx, pre, a, b=0):
    for k in ("origin", "val", "jac"):
        s = f"{pre}_{k.upper()}"
        if ctx.seq_size(s):
            ctx.upload(s, a[k][:ctx.seq_size(s)], b, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("k_scale", [1.0, 30.0])
def test_forward_parity_talos_full_horizon(gpu, k_scale):
    """forward_pass (ddp_fwd.ipp:9-67) on the Talos-like tree at the benchmark horizon T = 200, 8 step sizes per round, with
    and without a forced halving: same accepted step as the oracle's sequential halving, same summed cost difference, same
    trajectory.  200 closed-loop steps amplify the few-ulp difference between the two ABA restatements, so the whole
    trajectory is held to 1e-7 and its first 20 steps to 1e-9 (the T = 10 cases of test_dynamics_parity hold 1e-9)."""
    capi = gpu
    T = 200
    model, spec, o = make("tree38", T, fd_mode=0)
    xs, us, d, mults, bw = _one_iteration_inputs(o, model, 21, 1.0, 0.3, 0.0)
    bw["fb"]["val"] *= k_scale
    step_ref, xs_ref, us_ref, n_evals = o.forward(xs, us, mults, bw["fb"], bw["mu"])
    if k_scale > 1:
        assert n_evals > 1 and step_ref < 1.0, (step_ref, n_evals)
    nx, m = o.nx, o.m
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        assert ctx.info()["fwd_path"] == 1            # the latency path is the one the benchmark runs
        _upload_traj(ctx, xs, us)
        ctx.upload("X_NEW", xs, 0, 1); ctx.upload("U_NEW", us, 0, 1)
        _upload_affine(ctx, "FB", bw["fb"])
        rc, step, dcost = ctx.forward(bw["mu"], n_alpha=8)
        assert step[0] == step_ref, (step, step_ref, n_evals)
        xn = ctx.download("X_NEW", 0, 1)[0]
        un = ctx.download("U_NEW", 0, 1)[0]
        assert rel_err(xn[:21 * nx], xs_ref[:21 * nx]) < 1e-9 and rel_err(un[:20 * m], us_ref[:20 * m]) < 1e-9
        assert rel_err(xn, xs_ref) < 1e-7, rel_err(xn, xs_ref)
        assert rel_err(un, us_ref) < 1e-7, rel_err(un, us_ref)
        dc_ref, _, _ = o.forward_alpha(step_ref, xs, us, mults, bw["fb"], bw["mu"])
        assert dcost[0] <= 0 and abs(dcost[0] - dc_ref) <= 1e-9 * max(1.0, abs(dc_ref)), (dcost[0], dc_ref)


@pytest.mark.gpu
def test_frame_constrained_chain_full_horizon(gpu):
    """BASELINE config 5 at its size (test/pinocchio_spatial_eq_ddp.cpp:30-53 shape: 3-row frame translation at
    t = T-2 after two time shifts), T = 200: linearise (mode-2 tensors included) vs the oracle, then the constrained
    backward sweep and the forward sweep on identical derivative inputs (the device's own).

    The sweeps run tensor-free.  The reference takes the constraint jacobian from getFrameJacobian(..., WORLD)
    (pinocchio_model.ipp:458-461), which is not d(position)/dq; the mode-2 stencil (problem.hpp:192-222) subtracts
    eps * that column from a difference of the true function, so eq_xx / eq_uu carry 2 (J_world - J_true) / eps ~ 1e4 and
    Q_uu + reg I never becomes positive definite (both sides restart forever: checked on the oracle).  The reference's own
    driver for this shape uses mode 1 (dy{model, 0.01, false}); that configuration is covered in test_analytic_derivs."""
    capi = gpu
    T, mu = 200, 1e3
    model, spec, o = make("chain6_frame", T, fd_mode=2)
    x0, us, xs = held_trajectory(o, model, seed=5)          # a posture held for 2 s (a well-conditioned 200-step recursion)
    assert float(np.max(np.abs(xs))) < 10.0
    d_ref = o.compute_derivatives(xs, us)
    rng = np.random.default_rng(5)
    mults = o.alloc_affine(o.Etot)
    mults["origin"][:] = xs[:T * o.nx]
    mults["jac"][:o.Etot * o.n] = rng.normal(size=o.Etot * o.n)            # N(0,1) seed
    fscale = max(1.0, float(np.max(np.abs(d_ref["f_val"]))))
    tol1 = 8 * EPS * fscale / E1
    tol2 = 64 * EPS * fscale / (E2 * E2) + 4 * tol1 / E2
    with capi.Context(spec) as ctx:
        _upload_traj(ctx, xs, us)
        ctx.linearize()
        for key, seq in {**DERIV_SEQS, **TENSOR_SEQS}.items():
            sz = ctx.seq_size(seq)
            if not sz:
                continue
            got = ctx.download(seq, 0, 1)[0]
            err, scale = float(np.max(np.abs(got - d_ref[key][:sz]))), max(1.0, float(np.max(np.abs(d_ref[key][:sz]))))
            if key in ("lfx", "lfxx", "lx", "lu", "lxx", "lux", "luu"):
                assert err == 0.0, key
            elif key in ("f_val", "eq_val"):
                assert err <= 1e-12 * scale, (key, err)
            elif key in ("fx", "fu", "eq_x", "eq_u"):
                assert err <= tol1 * (4 if key.startswith("eq") else 1) * scale, (key, err)
            else:
                assert err <= tol2 * (8 if key.startswith("eq") else 1) * scale, (key, err, tol2)
    model, spec, o = make("chain6_frame", T, fd_mode=0)
    with capi.Context(spec, flags=capi.FLAG_TRACE | capi.FLAG_NO_TENSORS) as ctx:
        _upload_traj(ctx, xs, us)
        ctx.upload("X_NEW", xs, 0, 1); ctx.upload("U_NEW", us, 0, 1)
        _upload_affine(ctx, "MULT", mults)
        ctx.linearize()
        d = o.alloc_derivs()
        for key, seq in DERIV_SEQS.items():
            sz = ctx.seq_size(seq)
            if sz:
                d[key][:sz] = ctx.download(seq, 0, 1)[0]
        assert o.ne[T - 2] == 3 and float(np.max(np.abs(d["eq_x"][:3 * o.n]))) > 0          # the constraint really is there
        ref = o.backward(d, xs, mults, 0.0, mu)
        rc, reg, mu_out, restarts = ctx.backward(0.0, mu)
        assert restarts[0] == ref["restarts"] and mu_out[0] == ref["mu"] and reg[0] == ref["reg"]
        got = {seq: ctx.download(seq, 0, 1)[0] for seq in ("FB_VAL", "FB_JAC", "VX_TRACE", "VXX_TRACE")}
        # (a) every one of the 200 steps on its own, from the device's own V(t+1): the north star's 1e-10
        from oracle.binding import Oracle
        target = np.array([0.3, 0.2, 0.4])

        def one_step_oracle(t):
            e = int(o.ne[t])
            return Oracle(model, 1, dt=0.01, c=1.0, fd_mode=0, eq_kind=capi.EQ_FRAME, eq_advance=2, ne=np.array([e], dtype=np.int64),
                          eq_target=target[:e], frame_joint=5, frame_off=(0.0, 0.0, 0.0823))
        # (with the reg / mu the sweep ended on: the random multiplier jacobians make V_xx indefinite once and the sweep
        # restarts, ddp_bwd.ipp:105-110 -- the same restart on both sides, asserted above)
        assert restarts[0] >= 1
        worst = stepwise_backward_check(one_step_oracle, o, d, xs, mults, reg[0], mu_out[0], got["VX_TRACE"], got["VXX_TRACE"],
                                        got["FB_VAL"], got["FB_JAC"], range(T))
        assert worst < 1e-10, worst
        # (b) the whole recursion end to end.  With l = c/2 |u|^2 only and f_u ~ 125 (the light wrist), V_xx = Q_xx - Q_ux^T
        # Q_uu^-1 Q_ux cancels to ~1e-6 of its terms at every step: 200 steps carry one ulp to ~4e-7 between any two correct
        # double implementations (measured; the reference runs this shape in 1000-digit mpfr for that reason)
        for seq, r in (("FB_VAL", ref["fb"]["val"]), ("FB_JAC", ref["fb"]["jac"]), ("VX_TRACE", ref["Vx"]), ("VXX_TRACE", ref["Vxx"])):
            assert rel_err(got[seq], r[:got[seq].size]) < 1e-5, (seq, rel_err(got[seq], r[:got[seq].size]))
            tail = slice(-20 * (got[seq].size // T), None)                    # the first 20 steps of the sweep (t = 180 .. 199)
            assert rel_err(got[seq][tail], r[:got[seq].size][tail]) < 1e-10, seq
        # forward sweep on identical gains: the oracle's are uploaded (the two sets differ by the 4e-7 above)
        _upload_affine(ctx, "FB", ref["fb"])
        step_ref, xs_ref, us_ref, _ = o.forward(xs, us, mults, ref["fb"], ref["mu"])
        rc, step, dcost = ctx.forward(mu_out, n_alpha=8)
        assert step[0] == step_ref, (step, step_ref)
        assert rel_err(ctx.download("X_NEW", 0, 1)[0], xs_ref) < 1e-8
        assert rel_err(ctx.download("U_NEW", 0, 1)[0], us_ref) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("T,sweeps", [(200, False), (60, True)])
def test_free_floating_frame_constrained(gpu, T, sweeps):
    """BASELINE config 5 as worded: free-floating base (SE(3) root, nq = 39, nv = 38) + frame equality constraint, constrained
    Riccati on one GPU.  x is 77 wide, the sweep's matrices 76: the case exercises nx != n everywhere.

    T = 200: first-order linearisation on SE(3) (FD through integrate / difference) of all 200 steps against the oracle.
    T = 60: the same, then the constrained backward sweep -- every step against the oracle from the device's own V(t+1) (1e-10)
    and the whole recursion end to end -- and the line search on identical gains.
    Why the sweep is not run at T = 200: with l = c/2 |u|^2 alone, V_xx is only positive SEMI-definite (rank 3 behind the
    constraint at T - 2), and in double the Riccati recursion through this model's f_x lets the null directions drift:
    lambda_min(V_xx) = -3e-12, -9e-10, -3e-8, -3e-6, -0.4 after 10, 40, 70, 90, 99 steps (measured on the oracle, both with and
    without the free flyer), then Q_uu stops being positive definite for every reg (reg only enters Q_uu, ddp_bwd.ipp:102), and
    the reference's restart loop never ends.  The reference runs this shape in 500-digit mpfr (test/pinocchio_spatial_eq_ddp.cpp)."""
    capi = gpu
    mu = 1e3
    model, spec, o = make("tree38ff_frame", T, fd_mode=0)
    assert o.nx == 77 and o.n == 76 and model.nq == 39
    x0, us, xs = held_trajectory(o, model, seed=6, q0_sigma=0.2)
    assert float(np.max(np.abs(xs))) < 10.0
    d_ref = o.compute_derivatives(xs, us)
    rng = np.random.default_rng(6)
    mults = o.alloc_affine(o.Etot)
    mults["origin"][:] = xs[:T * o.nx]
    mults["jac"][:o.Etot * o.n] = 0.1 * rng.normal(size=o.Etot * o.n)
    # forward-difference noise: the rounding of f over eps.  Holding 70 kg against gravity takes |u| ~ 850 through an M^-1 of
    # ~ 170: the accelerations are differences of terms of size dt |M^-1| |u| (= max|f_u| max|u|), and that is what eps divides
    fscale = max(1.0, float(np.max(np.abs(d_ref["f_val"]))), float(np.max(np.abs(d_ref["fu"]))) * float(np.max(np.abs(us))))
    tol1 = 8 * EPS * fscale / E1
    with capi.Context(spec, flags=capi.FLAG_TRACE | capi.FLAG_NO_TENSORS) as ctx:
        _upload_traj(ctx, xs, us)
        ctx.upload("X_NEW", xs, 0, 1); ctx.upload("U_NEW", us, 0, 1)
        _upload_affine(ctx, "MULT", mults)
        ctx.linearize()
        d = o.alloc_derivs()
        for key, seq in DERIV_SEQS.items():
            sz = ctx.seq_size(seq)
            if not sz:
                continue
            got = ctx.download(seq, 0, 1)[0]
            d[key][:sz] = got
            err, scale = float(np.max(np.abs(got - d_ref[key][:sz]))), max(1.0, float(np.max(np.abs(d_ref[key][:sz]))))
            if key in ("lfx", "lx", "lu"):
                assert err == 0.0, key
            elif key in ("f_val", "eq_val"):
                assert err <= 1e-12 * scale, (key, err)
            else:
                assert err <= tol1 * (4 if key.startswith("eq") else 1), (key, err, tol1)
        assert o.ne[T - 2] == 3 and float(np.max(np.abs(d["eq_x"][:3 * o.n]))) > 0
        if not sweeps:
            return
        ref = o.backward(d, xs, mults, 0.0, mu)
        assert ref["restarts"] >= 0
        rc, reg, mu_out, restarts = ctx.backward(0.0, mu)
        assert restarts[0] == ref["restarts"] and mu_out[0] == ref["mu"] and reg[0] == ref["reg"]
        got = {seq: ctx.download(seq, 0, 1)[0] for seq in ("FB_VAL", "FB_JAC", "VX_TRACE", "VXX_TRACE")}
        from oracle.binding import Oracle
        target = np.array([0.4, -0.1, 0.9])

        def one_step_oracle(t):
            e = int(o.ne[t])
            return Oracle(model, 1, dt=0.01, c=1.0, fd_mode=0, eq_kind=capi.EQ_FRAME, eq_advance=2, ne=np.array([e], dtype=np.int64),
                          eq_target=target[:e], frame_joint=22, frame_off=(0.0, 0.0, 0.1))
        worst = stepwise_backward_check(one_step_oracle, o, d, xs, mults, reg[0], mu_out[0], got["VX_TRACE"], got["VXX_TRACE"],
                                        got["FB_VAL"], got["FB_JAC"], range(T))
        assert worst < 1e-10, worst
        for seq, r in (("FB_VAL", ref["fb"]["val"]), ("FB_JAC", ref["fb"]["jac"]), ("VX_TRACE", ref["Vx"]), ("VXX_TRACE", ref["Vxx"])):
            assert rel_err(got[seq], r[:got[seq].size]) < 1e-6, (seq, rel_err(got[seq], r[:got[seq].size]))
        _upload_affine(ctx, "FB", ref["fb"])
        step_ref, xs_ref, us_ref, _ = o.forward(xs, us, mults, ref["fb"], ref["mu"])
        rc, step, dcost = ctx.forward(mu_out, n_alpha=8)
        assert step[0] == step_ref, (step, step_ref)
        assert rel_err(ctx.download("X_NEW", 0, 1)[0], xs_ref) < 1e-8
        assert rel_err(ctx.download("U_NEW", 0, 1)[0], us_ref) < 1e-8


@pytest.mark.gpu
def test_talos_linearize_full_horizon_vs_oracle_across_slices(gpu):
    """The static-topology stencil at T = 200 with more (instance, t) pairs than one workspace slice (1 024) holds,
    against the ORACLE (not the generic kernels) at the pairs that straddle the slices: first / last of slice 0, first of
    slice 1, the very last pair, and two in the middle."""
    capi = gpu
    T, B = 200, 6
    model, spec, o = make("tree38", T, batch=B, fd_mode=2)
    # every instance holds a random posture for 2 s under computed-torque control + noise: velocities stay O(1), so the
    # intermediates of the ABA stay at the scale of f and the finite-difference noise bound (a few ulp of f over eps) holds
    # at every t (on a free fall the late states carry |v| ~ 50 rad/s and the same comparison needs 40 ulp: measured)
    trajs = [held_trajectory(o, model, seed=70 + b, u_sigma=0.3) for b in range(B)]
    us = np.stack([tr[1] for tr in trajs])
    picks = [(0, 0), (0, 199), (2, 100), (5, 23), (5, 24), (5, 199)]        # pair index b T + t: 1 023 is (5, 23), 1 024 is (5, 24)
    n, m, nx = 2 * model.nv, model.nv, 2 * model.nv
    with capi.Context(spec) as ctx:
        assert ctx.info()["lin_path"] == 2                                  # static TopoTalos38
        ctx.upload("X", np.stack([tr[2] for tr in trajs]))
        ctx.upload("U", us)
        xs = ctx.download("X")
        ctx.linearize()
        got = {}
        for b in sorted({b for b, _ in picks}):
            for key, seq in (("f_val", "F_VAL"), ("fx", "FX"), ("fu", "FU"), ("fxx", "FXX"), ("fux", "FUX"), ("fuu", "FUU")):
                got[key, b] = ctx.download(seq, b, 1)[0]
    # the oracle evaluates compute_derivatives point by point: hand it the picked (x_t, u_t) as a short trajectory
    P = len(picks)
    _, _, op = make("tree38", P, fd_mode=2)
    xs_p = np.zeros((P + 1) * nx); us_p = np.zeros(P * m)
    for i, (b, t) in enumerate(picks):
        xs_p[i * nx:(i + 1) * nx] = xs[b, t * nx:(t + 1) * nx]
        us_p[i * m:(i + 1) * m] = us[b, t * m:(t + 1) * m]
    d = op.compute_derivatives(xs_p, us_p)
    fscale = max(1.0, float(np.max(np.abs(d["f_val"]))))
    tol1 = 8 * EPS * fscale / E1
    tol2 = 64 * EPS * fscale / (E2 * E2) + 4 * tol1 / E2
    sizes = {"f_val": nx, "fx": n * n, "fu": n * m, "fxx": n ** 3, "fux": n * m * n, "fuu": n * m * m}
    for i, (b, t) in enumerate(picks):
        for key, sz in sizes.items():
            a = got[key, b][t * sz:(t + 1) * sz]
            r = d[key][i * sz:(i + 1) * sz]
            err, scale = float(np.max(np.abs(a - r))), max(1.0, float(np.max(np.abs(r))))
            tol = 1e-12 if key == "f_val" else (tol1 if key in ("fx", "fu") else tol2)
            assert np.all(np.isfinite(a)) and err <= tol * scale, (key, b, t, err, tol * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,fd_mode,iters,thr,mu,w,n,u_sigma,tol", [
    ("pendulum", 50, 2, 12, 1e-8, 10.0, 1.0, 10.0, (0.0, 0.4), 1e-6),
    # instance 0 reaches its optimum at iteration 8, instance 1 is still going at 12 (checked on the oracle): the latch
    # (FD jacobians, mode 2: the finite-difference noise compounds over the iterations -- 1.2e-4 on the unconverged
    # instance, measured; the analytic-derivative configuration of the same problem holds 1e-8: test_analytic_derivs.py)
    ("chain6", 10, 2, 12, 1e-6, 1e4, 1e-1, 10.0, (0.01, 0.02), 1e-3),
])
def test_whole_solve_two_different_instances(gpu, name, T, fd_mode, iters, thr, mu, w, n, u_sigma, tol):
    """solve<M> (ddp.hpp:745-842) of a batch of two DIFFERENT problems: each instance must reproduce its own oracle solve
    (result, iteration count, mu, trajectory) whatever its batch-mate does -- an instance that reaches its optimum is
    returned as it is at that moment (ddp.hpp:799-800), not carried on.  ddp_hip_solve and the call-by-call Python loop
    must agree bit for bit."""
    from ddp_pinocchio_amd import solver
    capi = gpu
    B = 2
    model, spec, o = make(name, T, batch=B, fd_mode=fd_mode)
    rng = np.random.default_rng(3)
    seed = 0.01 * rng.normal(size=max(o.Etot * o.n, 1))
    x0 = np.zeros(2 * model.nv)
    inits, refs = [], []
    for b in range(B):
        us0 = u_sigma[b] * np.random.default_rng(40 + b).normal(size=T * model.nv)
        xs0 = o.rollout(x0, us0)
        inits.append((xs0, us0))
        refs.append(o.solve(xs0, us0, seed, max_iterations=iters, threshold=thr, mu=mu, reg=0.0, w=w, n=n))
    out = {}
    for which in ("native", "stepwise"):
        with capi.Context(spec, flags=0 if fd_mode else capi.FLAG_NO_TENSORS) as ctx:
            for b, (xs0, us0) in enumerate(inits):
                ctx.upload("X", xs0, b, 1); ctx.upload("U", us0, b, 1)
                ctx.upload("X_NEW", xs0, b, 1); ctx.upload("U_NEW", us0, b, 1)
                ctx.upload("MULT_ORIGIN", xs0[:T * o.nx], b, 1)
                if o.Etot:
                    ctx.upload("MULT_VAL", np.zeros(o.Etot), b, 1)
                    ctx.upload("MULT_JAC", seed[:o.Etot * o.n], b, 1)
            fn = solver.solve if which == "native" else solver.solve_stepwise
            log = fn(ctx, iters, thr, mu, 0.0, w, n)
            out[which] = (log, ctx.download("X"), ctx.download("U"))
    log, xs, us = out["native"]
    log2, xs2, us2 = out["stepwise"]
    assert np.array_equal(xs, xs2) and np.array_equal(us, us2)
    for k in ("iterations", "done", "mu", "reg", "w", "n", "opt_obj", "opt_constr"):
        assert np.array_equal(np.asarray(log[k]), np.asarray(log2[k])), k
    for b in range(B):
        xs_ref, us_ref, fb_ref, log_ref = refs[b]
        assert int(log["iterations"][b]) == log_ref["iterations"], (b, log["iterations"], log_ref["iterations"])
        assert bool(log["done"][b]) == bool(log_ref["result"] == 1), (b, log["done"], log_ref["result"])
        assert log["mu"][b] == log_ref["mu"], (b, log["mu"], log_ref["mu"])
        # absolute on states of O(1) rad (the optimum of the chain problem IS x = 0: a relative measure is meaningless)
        ex, eu = float(np.max(np.abs(xs[b] - xs_ref))), float(np.max(np.abs(us[b] - us_ref)))
        assert ex < tol * max(1.0, float(np.max(np.abs(xs_ref)))), (b, ex)
        assert eu < 10 * tol * max(1.0, float(np.max(np.abs(us_ref)))), (b, eu)
        assert abs(log["opt_constr"][b] - log_ref["opt_constr"]) <= 100 * tol * max(1.0, log_ref["opt_constr"])


@pytest.mark.gpu
def test_frozen_instance_is_left_alone(gpu):
    """ddp_hip_set_active: the sweeps skip an inactive instance and swap_traj keeps its trajectory."""
    capi = gpu
    T, B = 8, 3
    model, spec, o = make("chain6", T, batch=B, fd_mode=0)
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        for b in range(B):
            x0, us, xs = initial_trajectory(o, model, seed=60 + b, u_sigma=0.05)
            ctx.upload("X", xs, b, 1); ctx.upload("U", us, b, 1); ctx.upload("X_NEW", xs, b, 1); ctx.upload("U_NEW", us, b, 1)
            ctx.upload("MULT_ORIGIN", xs[:T * o.nx], b, 1)
        ctx.fill("MULT_VAL", 0.0); ctx.fill("MULT_JAC", 0.0)
        ctx.linearize()
        ctx.backward(0.0, 100.0)
        fb0 = ctx.download("FB_JAC"); x_before = ctx.download("X"); u_before = ctx.download("U")
        ctx.fill("FB_JAC", 7.0); ctx.fill("FB_VAL", 7.0)
        ctx.set_active([1, 0, 1])
        rc, reg, mu, restarts = ctx.backward(0.0, 100.0)
        fb = ctx.download("FB_JAC")
        assert np.array_equal(fb[0], fb0[0]) and np.array_equal(fb[2], fb0[2]) and np.all(fb[1] == 7.0)
        ctx.upload("FB_JAC", fb0[1], 1, 1)
        rc, step, _ = ctx.forward(mu, n_alpha=8)
        ctx.swap_traj()
        x_after = ctx.download("X"); u_after = ctx.download("U")
        assert np.array_equal(x_after[1], x_before[1]) and np.array_equal(u_after[1], u_before[1])
        assert not np.array_equal(u_after[0], u_before[0]) and not np.array_equal(u_after[2], u_before[2])
        ctx.set_active(None)
