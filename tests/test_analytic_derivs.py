"""Analytic ABA derivatives on the device (reference: first_order_deriv problem.hpp:463-503 with
model_t::d_dynamics_aba pinocchio_model.ipp:359-400) and FD mode 1 on tree models (problem.hpp:67-150: forward
differences of the analytic jacobians -- what both reference UR5 drivers use, test/pinocchio_ddp.cpp:60).

f_x / f_u are held to 1e-10 relative (no finite-difference noise any more: the two sides differ by the rounding of two
independent restatements of the same recursion).  The mode-1 tensors are forward differences of those jacobians with
eps = 1.5e-8: a few ulp of the jacobian entries over eps."""
import numpy as np
import pytest

from problems import held_trajectory, initial_trajectory, make
from synth import rel_err
from test_dynamics_parity import DERIV_SEQS, TENSOR_SEQS, _upload_traj

EPS, E1 = 2.220446049250313e-16, 1.4901161193847656e-08


def _check_linearize(capi, name, T, traj, ulps=8, batch=1, fd_mode=1):
    model, spec, o = make(name, T, fd_mode=fd_mode, first_order_fd=0, batch=batch)
    x0, us, xs = traj(o, model)
    d = o.compute_derivatives(xs, us)
    with capi.Context(spec) as ctx:
        assert ctx.info()["first_order"] == 2
        for b in range(batch):
            _upload_traj(ctx, xs, us, b)
        ctx.linearize()
        jscale = max(1.0, float(np.max(np.abs(d["fx"]))), float(np.max(np.abs(d["fu"]))))
        # the entries of M^-1 (f_u = dt M^-1, d qdd/dx = -M^-1 d tau/dx) are only good to cond(M) ulps on either side, and the
        # forward difference divides that by eps: the noise floor of mode 1 in double (the reference runs it in mpfr)
        nv = model.nv
        cond = max(float(np.linalg.cond(o.crba(xs[t * 2 * nv:t * 2 * nv + nv]))) for t in range(0, T, max(1, T // 8)))
        for b in range(batch):
            for key, seq in {**DERIV_SEQS, **TENSOR_SEQS}.items():
                sz = ctx.seq_size(seq)
                if not sz:
                    continue
                got = ctx.download(seq, b, 1)[0]
                ref = d[key][:sz]
                err, scale = float(np.max(np.abs(got - ref))), max(1.0, float(np.max(np.abs(ref))))
                assert np.all(np.isfinite(got)), key
                if key in ("lfx", "lfxx", "lx", "lu", "lxx", "lux", "luu"):
                    assert err == 0.0, key
                elif key in ("f_val", "eq_val"):
                    assert err <= 1e-12 * scale, (key, err)
                elif key in ("fx", "fu", "eq_x", "eq_u"):
                    assert err <= 1e-10 * scale, (key, err, scale)                       # analytic: the north star's 1e-10
                elif fd_mode != 1:
                    continue                                                            # mode-2 tensors: test_dynamics_parity.py's FD-noise bounds
                elif key in ("fxx", "fux", "fuu"):
                    assert err <= ulps * EPS * cond * jscale / E1, (key, err, ulps * EPS * cond * jscale / E1)
                else:                                                                   # eq tensors: two chained linearisations
                    assert err <= 8 * ulps * EPS * cond * jscale * scale / E1, (key, err)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("name,T", [("chain6", 10), ("chain6_frame", 10), ("chain6", 100)])
def test_analytic_linearize_chain(gpu, name, T):
    """the UR5-like chain in the derivative mode of the reference's own drivers: dy{model, 0.01, false}"""
    _check_linearize(gpu, name, T, lambda o, model: held_trajectory(o, model, seed=3, u_sigma=0.05))


@pytest.mark.gpu
def test_mode1_structure_zero_blocks(gpu):
    """M^-1 depends on q alone, so f_u is the same at (x, u), (x + eps e_v, u) and (x, u + eps e_u): the v-slabs of f_ux
    and all of f_uu are exactly zero, as they are in the reference (fu_ - fu == 0 before the division by eps)"""
    capi = gpu
    T = 4
    model, spec, o = make("chain6", T, fd_mode=1, first_order_fd=0)
    x0, us, xs = held_trajectory(o, model, seed=4, u_sigma=0.05)
    with capi.Context(spec) as ctx:
        _upload_traj(ctx, xs, us)
        ctx.linearize()
        n, m, nv = o.n, o.m, model.nv
        fuu = ctx.download("FUU", 0, 1)[0]
        fux = ctx.download("FUX", 0, 1)[0].reshape(T, n, m, n)      # [t][k = x index][j][i]
        assert np.all(fuu == 0.0)
        assert np.all(fux[:, nv:, :, :] == 0.0) and np.any(fux[:, :nv, :, :] != 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("fd_mode,tol", [(0, 1e-8), (1, 2e-2)])
def test_analytic_whole_solve_chain(gpu, fd_mode, tol):
    """test/pinocchio_ddp.cpp's configuration (UR5-like chain, config constraint every step, analytic first order) end to
    end on a batch of two different instances, against each instance's own oracle solve.
    fd_mode 0 (tensor-free): no finite difference anywhere -- the final trajectories agree to 1e-8 (round 1, FD jacobians:
    1e-4).  fd_mode 1 (the driver's dy{model, 0.01, false}): the tensors are forward differences of jacobians that are good
    to cond(M) ulps, i.e. they carry ~1e-3 relative noise in double on either side (the oracle run twice with its input
    perturbed by 1e-13 moves its own answer by 1e-5); logs must agree, the trajectories to that noise."""
    from ddp_pinocchio_amd import solver
    capi = gpu
    T, B, iters, thr, mu, w, n = 10, 2, 12, 1e-6, 1e4, 1e-1, 10.0
    model, spec, o = make("chain6", T, batch=B, fd_mode=fd_mode, first_order_fd=0)
    rng = np.random.default_rng(3)
    seed = 0.01 * rng.normal(size=o.Etot * o.n)
    inits, refs = [], []
    for b, sig in enumerate((0.01, 0.02)):
        us0 = sig * np.random.default_rng(40 + b).normal(size=T * model.nv)
        xs0 = o.rollout(np.zeros(2 * model.nv), us0)
        inits.append((xs0, us0))
        refs.append(o.solve(xs0, us0, seed, max_iterations=iters, threshold=thr, mu=mu, reg=0.0, w=w, n=n))
    with capi.Context(spec, flags=0 if fd_mode else capi.FLAG_NO_TENSORS) as ctx:
        for b, (xs0, us0) in enumerate(inits):
            ctx.upload("X", xs0, b, 1); ctx.upload("U", us0, b, 1); ctx.upload("X_NEW", xs0, b, 1); ctx.upload("U_NEW", us0, b, 1)
            ctx.upload("MULT_ORIGIN", xs0[:T * o.nx], b, 1)
            ctx.upload("MULT_VAL", np.zeros(o.Etot), b, 1)
            ctx.upload("MULT_JAC", seed, b, 1)
        log = solver.solve(ctx, iters, thr, mu, 0.0, w, n)
        xs, us = ctx.download("X"), ctx.download("U")
    for b in range(B):
        xs_ref, us_ref, fb_ref, log_ref = refs[b]
        assert int(log["iterations"][b]) == log_ref["iterations"] and bool(log["done"][b]) == bool(log_ref["result"] == 1)
        assert log["mu"][b] == log_ref["mu"] and log["reg"][b] == log_ref["reg"]
        assert float(np.max(np.abs(xs[b] - xs_ref))) < tol, (b, float(np.max(np.abs(xs[b] - xs_ref))))
        assert float(np.max(np.abs(us[b] - us_ref))) < 100 * tol * max(1.0, float(np.max(np.abs(us_ref)))), b
        assert abs(log["opt_constr"][b] - log_ref["opt_constr"]) <= 100 * tol * max(1.0, log_ref["opt_constr"])


@pytest.mark.gpu
@pytest.mark.parametrize("T", [3, 200])
def test_analytic_linearize_talos(gpu, T):
    """the Talos-like tree: wave-per-evaluation kernels (lin_analytic.hip).  T = 200 x 6 instances exercises the workspace
    slices; the oracle is evaluated at picked (instance, t) pairs"""
    capi = gpu
    if T == 3:
        _check_linearize(capi, "tree38", T, lambda o, model: held_trajectory(o, model, seed=9, u_sigma=0.3))
        return
    B = 6
    model, spec, o = make("tree38", T, batch=B, fd_mode=1, first_order_fd=0)
    trajs = [held_trajectory(o, model, seed=70 + b, u_sigma=0.3) for b in range(B)]
    picks = [(0, 0), (0, 199), (2, 100), (5, 23), (5, 24), (5, 199)]
    n, m, nx = o.n, o.m, o.nx
    with capi.Context(spec) as ctx:
        assert ctx.info()["lin_path"] == 2      # the perturbed points' accelerations come from the static first-order kernels
        ctx.upload("X", np.stack([tr[2] for tr in trajs])); ctx.upload("U", np.stack([tr[1] for tr in trajs]))
        ctx.linearize()
        got = {(k, b): ctx.download(s, b, 1)[0] for b in sorted({b for b, _ in picks})
               for k, s in (("f_val", "F_VAL"), ("fx", "FX"), ("fu", "FU"), ("fxx", "FXX"), ("fux", "FUX"), ("fuu", "FUU"))}
    P = len(picks)
    _, _, op = make("tree38", P, fd_mode=1, first_order_fd=0)
    xs_p = np.zeros((P + 1) * nx); us_p = np.zeros(P * m)
    for i, (b, t) in enumerate(picks):
        xs_p[i * nx:(i + 1) * nx] = trajs[b][2][t * nx:(t + 1) * nx]
        us_p[i * m:(i + 1) * m] = trajs[b][1][t * m:(t + 1) * m]
    d = op.compute_derivatives(xs_p, us_p)
    jscale = max(1.0, float(np.max(np.abs(d["fx"]))), float(np.max(np.abs(d["fu"]))))
    cond = max(float(np.linalg.cond(op.crba(xs_p[i * nx:i * nx + model.nv]))) for i in range(P))
    sizes = {"f_val": nx, "fx": n * n, "fu": n * m, "fxx": n ** 3, "fux": n * m * n, "fuu": n * m * m}
    for i, (b, t) in enumerate(picks):
        for key, sz in sizes.items():
            a, r = got[key, b][t * sz:(t + 1) * sz], d[key][i * sz:(i + 1) * sz]
            err, scale = float(np.max(np.abs(a - r))), max(1.0, float(np.max(np.abs(r))))
            tol = 1e-12 * scale if key == "f_val" else (1e-10 * scale if key in ("fx", "fu") else 8 * EPS * cond * jscale / E1)
            assert np.all(np.isfinite(a)) and err <= tol, (key, b, t, err, tol)


@pytest.mark.gpu
def test_mode1_static_accelerations_against_own_forward_dynamics(gpu, monkeypatch):
    """mode 1 on a tree with a static topology takes the accelerations of its 2 nv perturbed points from the static first-order
    kernels (lin_static.hip level 6) instead of one cooperative ABA per point (DDP_HIP_ANA_OWN_ABA=1 keeps the latter): two
    roundings of the same accelerations, so the tensors agree to a few ulp of the jacobians over eps, f_x to rounding and f_u bit for bit"""
    capi = gpu
    T, B = 5, 2
    model, spec, o = make("tree38", T, batch=B, fd_mode=1, first_order_fd=0)
    trajs = [held_trajectory(o, model, seed=31 + b, u_sigma=0.3) for b in range(B)]
    out = {}
    for own in (False, True):
        if own:
            monkeypatch.setenv("DDP_HIP_ANA_OWN_ABA", "1")
        with capi.Context(spec) as ctx:
            assert ctx.info()["lin_path"] == (1 if own else 2)
            ctx.upload("X", np.stack([tr[2] for tr in trajs])); ctx.upload("U", np.stack([tr[1] for tr in trajs]))
            ctx.linearize()
            out[own] = {k: ctx.download(k, 0, B) for k in ("FX", "FU", "FXX", "FUX", "FUU")}
    nv = model.nv
    cond = max(float(np.linalg.cond(o.crba(trajs[0][2][t * 2 * nv:t * 2 * nv + nv]))) for t in range(T))
    jscale = max(1.0, float(np.max(np.abs(out[True]["FX"]))), float(np.max(np.abs(out[True]["FU"]))))
    # (the trajectory point's own acceleration comes from the static kernels too: f_x differs by its rounding, f_u = dt M^-1 does not)
    assert rel_err(out[False]["FX"], out[True]["FX"]) < 1e-12 and np.array_equal(out[False]["FU"], out[True]["FU"])
    assert np.array_equal(out[False]["FUU"], out[True]["FUU"])
    for k in ("FXX", "FUX"):
        err = float(np.max(np.abs(out[False][k] - out[True][k])))
        assert err <= 8 * EPS * cond * jscale / E1, (k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [1, 2])
def test_model_point_evaluations(gpu, which):
    """ddp_hip_model_* (seam B2: what a host-side model_t<double> calls, adapters/pinocchio_double.cpp) against the oracle:
    dynamics_aba, d_dynamics_aba, frame_coordinates / d_frame_coordinates at a random configuration"""
    capi = gpu
    from oracle.binding import Oracle
    bm = capi.BuiltinModel(which, 1)
    o = Oracle(bm, 1)
    rng = np.random.default_rng(12)
    N = bm.nv
    q, v, tau = rng.normal(size=N), rng.normal(size=N), 2.0 * rng.normal(size=N)
    with capi.ModelHandle(bm) as h:
        assert rel_err(h.aba(q, v, tau), o.aba(q, v, tau)) < 1e-12
        got, ref = h.aba_derivatives(q, v, tau), o.aba_derivatives(q, v, tau)
        for g, r in zip(got, ref):
            assert rel_err(g, r) < 1e-10
        joint, off = N - 1, np.array([0.01, -0.02, 0.08])
        p3, J = h.frame(joint, off, q)
        assert rel_err(p3, o.frame_position(joint, off, q)) < 1e-13
        assert float(np.max(np.abs(J - o.frame_jacobian(joint, off, q)))) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,fd_mode", [("tree38_config", 3, 1), ("tree38_frame", 4, 1), ("tree38_config", 3, 2), ("tree38_frame", 4, 0)])
def test_analytic_linearize_talos_constrained(gpu, name, T, fd_mode):
    """What adapters/ddp_hip_bridge.hpp asks for on every reference driver -- analytic first order (problem.hpp:463-503),
    FD mode 1 (dy{model, 0.01, false}, test/pinocchio_ddp.cpp:60) and a constraint under two constraint_advance_time_t
    wrappers -- at the Talos size (round 2 refused it: lin.hip:849,896).  The chain rule runs on the analytic jacobians
    (lin_analytic.hip: ana_eq_kernel), its mode-1 tensors are forward differences of the chained jacobians
    (problem.hpp:611-620).  eq_x / eq_u to 1e-10, tensors at the cond(M) noise floor of mode 1 in double."""
    _check_linearize(gpu, name, T, lambda o, model: held_trajectory(o, model, seed=13, u_sigma=0.3), fd_mode=fd_mode)


@pytest.mark.gpu
def test_analytic_linearize_talos_constrained_slices(gpu):
    """the same over more (instance, t) pairs than one workspace slice holds (DDP_HIP_ANA_BT pairs, 256 by default):
    2 instances x T = 150; the oracle is evaluated at picked pairs, one of them the constrained step of the frame problem"""
    capi = gpu
    T, B = 150, 2
    model, spec, o = make("tree38_frame", T, batch=B, fd_mode=1, first_order_fd=0)
    trajs = [held_trajectory(o, model, seed=90 + b, u_sigma=0.3) for b in range(B)]
    n, m, nx, e = o.n, o.m, o.nx, 3
    with capi.Context(spec) as ctx:
        assert ctx.info()["lin_path"] == 2      # the perturbed points' accelerations come from the static first-order kernels
        ctx.upload("X", np.stack([tr[2] for tr in trajs])); ctx.upload("U", np.stack([tr[1] for tr in trajs]))
        ctx.linearize()
        got = {(k, b): ctx.download(sq, b, 1)[0] for b in range(B)
               for k, sq in (("eq_val", "EQ_VAL"), ("eq_x", "EQ_X"), ("eq_u", "EQ_U"), ("eq_xx", "EQ_XX"), ("eq_ux", "EQ_UX"), ("eq_uu", "EQ_UU"),
                             ("fx", "FX"), ("fxx", "FXX"))}
    # oracle on a 2-step problem whose constrained step (t = T' - 2 = 0) is the picked pair
    for b in range(B):
        _, _, op = make("tree38_frame", 2, fd_mode=1, first_order_fd=0)
        t = T - 2
        xs_p = np.zeros(3 * nx); us_p = np.zeros(2 * m)
        xs_p[:nx] = trajs[b][2][t * nx:(t + 1) * nx]; us_p[:m] = trajs[b][1][t * m:(t + 1) * m]
        xs_p[nx:2 * nx] = trajs[b][2][(t + 1) * nx:(t + 2) * nx]; us_p[m:] = trajs[b][1][(t + 1) * m:(t + 2) * m]
        d = op.compute_derivatives(xs_p, us_p)
        jscale = max(1.0, float(np.max(np.abs(d["fx"]))), float(np.max(np.abs(d["fu"]))))
        cond = float(np.linalg.cond(op.crba(xs_p[:model.nv])))
        for key, sz in (("eq_val", e), ("eq_x", e * n), ("eq_u", e * m), ("eq_xx", e * n * n), ("eq_ux", e * m * n), ("eq_uu", e * m * m)):
            a, r = got[key, b][:sz], d[key][:sz]
            err, scale = float(np.max(np.abs(a - r))), max(1.0, float(np.max(np.abs(r))))
            tol = 1e-12 * scale if key == "eq_val" else (1e-10 * scale if key in ("eq_x", "eq_u") else 64 * EPS * cond * jscale * scale / E1)
            assert np.all(np.isfinite(a)) and err <= tol, (key, b, err, tol)
        a, r = got["fx", b][t * n * n:(t + 1) * n * n], d["fx"][:n * n]
        assert float(np.max(np.abs(a - r))) <= 1e-10 * max(1.0, float(np.max(np.abs(r))))
        a, r = got["fxx", b][t * n ** 3:(t + 1) * n ** 3], d["fxx"][:n ** 3]
        assert float(np.max(np.abs(a - r))) <= 8 * EPS * cond * jscale / E1


@pytest.mark.gpu
def test_analytic_whole_solve_talos_drivers_mode(gpu):
    """test/pinocchio_ddp.cpp's exact configuration at the Talos size: config constraint on every joint at every step under
    two constraint_advance_time_t wrappers, dy{model, 0.01, false} = analytic first order + FD mode 1 -- what the bridge
    (adapters/ddp_hip_bridge.hpp) requests.  Logs (iterations, mu, reg) must agree with the oracle's solve; trajectories to the
    mode-1 noise (see test_analytic_whole_solve_chain)."""
    from ddp_pinocchio_amd import solver
    capi = gpu
    T, iters, thr, mu, w, n = 6, 5, 1e-6, 1e4, 1e-1, 10.0
    model, spec, o = make("tree38_config", T, batch=1, fd_mode=1, first_order_fd=0)
    rng = np.random.default_rng(5)
    seed = 0.01 * rng.normal(size=o.Etot * o.n)
    us0 = 0.01 * np.random.default_rng(41).normal(size=T * model.nv)
    xs0 = o.rollout(np.zeros(2 * model.nv), us0)
    xs_ref, us_ref, fb_ref, log_ref = o.solve(xs0, us0, seed, max_iterations=iters, threshold=thr, mu=mu, reg=0.0, w=w, n=n)
    with capi.Context(spec) as ctx:
        assert ctx.info()["first_order"] == 2
        ctx.upload("X", xs0); ctx.upload("U", us0); ctx.upload("X_NEW", xs0); ctx.upload("U_NEW", us0)
        ctx.upload("MULT_ORIGIN", xs0[:T * o.nx]); ctx.upload("MULT_VAL", np.zeros(o.Etot)); ctx.upload("MULT_JAC", seed)
        log = solver.solve(ctx, iters, thr, mu, 0.0, w, n)
        xs, us = ctx.download("X")[0], ctx.download("U")[0]
    assert int(log["iterations"][0]) == log_ref["iterations"] and bool(log["done"][0]) == bool(log_ref["result"] == 1)
    assert log["mu"][0] == log_ref["mu"] and log["reg"][0] == log_ref["reg"]
    assert float(np.max(np.abs(xs - xs_ref))) < 2e-2, float(np.max(np.abs(xs - xs_ref)))
    assert float(np.max(np.abs(us - us_ref))) < 2.0 * max(1.0, float(np.max(np.abs(us_ref))))


@pytest.mark.gpu
@pytest.mark.parametrize("name,T", [("tree38_config", 3), ("tree38_frame", 4), ("tree38", 3)])
def test_fused_evaluation_kernel_against_the_three_kernel_form(gpu, monkeypatch, name, T):
    """lin_analytic.hip has ONE arithmetic in three arrangements: the fused kernel (an evaluation goes from the state to its tensor
    slabs inside one wave; the config constraint's tensors come out of the same wave), the fused kernel with the constraint
    chain in its own kernel (DDP_HIP_ANA_EQ_KERNEL=1, what the frame constraint uses), and the three-kernel form with HBM
    workspaces (DDP_HIP_ANA_SPLIT=1).  Every output must agree bit for bit."""
    capi = gpu
    model, spec, o = make(name, T, batch=2, fd_mode=1, first_order_fd=0)
    trajs = [held_trajectory(o, model, seed=5 + b, u_sigma=0.3) for b in range(2)]
    keys = ("FX", "FU", "FXX", "FUX", "FUU", "EQ_VAL", "EQ_X", "EQ_U", "EQ_XX", "EQ_UX", "EQ_UU")

    def run():
        with capi.Context(spec) as ctx:
            ctx.upload("X", np.stack([tr[2] for tr in trajs])); ctx.upload("U", np.stack([tr[1] for tr in trajs]))
            ctx.linearize()
            return {k: ctx.download(k, 0, 2) for k in keys if ctx.seq_size(k)}
    fused = run()
    monkeypatch.setenv("DDP_HIP_ANA_EQ_KERNEL", "1")
    eqk = run()
    monkeypatch.delenv("DDP_HIP_ANA_EQ_KERNEL")
    monkeypatch.setenv("DDP_HIP_ANA_SPLIT", "1")
    split = run()
    monkeypatch.delenv("DDP_HIP_ANA_SPLIT")
    for k in fused:
        assert np.all(np.isfinite(fused[k])), k
        assert np.array_equal(fused[k], eqk[k]), (k, "constraint chain in its own kernel")
        assert np.array_equal(fused[k], split[k]), (k, "three-kernel form")
    if "EQ_UU" in fused and name == "tree38_config":
        assert not np.any(fused["EQ_UU"]) and float(np.max(np.abs(fused["EQ_XX"]))) > 0


@pytest.mark.gpu
def test_mode1_linearise_is_reproducible_run_to_run(gpu):
    """The fused evaluation kernel orders its LDS traffic by program order alone (its work-group is one wave: no s_barrier) and
    keeps prefetched global reads in flight across phases.  A race would show as run-to-run differences: eight linearisations of
    the same (T = 40, 3 instances, constrained) problem must give the same bits every time."""
    capi = gpu
    T, B = 40, 3
    model, spec, o = make("tree38_config", T, batch=B, fd_mode=1, first_order_fd=0)
    rng = np.random.default_rng(11)
    us = 0.1 * rng.normal(size=(B, T * o.m))
    with capi.Context(spec) as ctx:
        ctx.upload("X", np.zeros((B, (T + 1) * o.nx))); ctx.upload("U", us); ctx.rollout()
        ref = None
        for rep in range(8):
            ctx.fill("FXX", float("nan")); ctx.fill("FUX", float("nan")); ctx.fill("EQ_XX", float("nan")); ctx.fill("EQ_UX", float("nan"))
            ctx.linearize()
            got = {k: ctx.download(k) for k in ("FX", "FU", "FXX", "FUX", "FUU", "EQ_X", "EQ_U", "EQ_XX", "EQ_UX", "EQ_UU")}
            assert all(np.all(np.isfinite(v)) for v in got.values())
            if ref is None:
                ref = got
            else:
                for k in got:
                    assert np.array_equal(got[k], ref[k]), (k, rep)
