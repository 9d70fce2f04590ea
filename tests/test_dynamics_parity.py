"""Rollout / linearisation / forward-sweep parity: HIP path (C-ABI) vs the CPU oracle.

Tolerances.  The rigid-body arithmetic is restated independently on both sides (structured 3x3 block
transforms + FMA contraction + ocml sincos on the GPU, dense 6x6 products + no contraction + glibc on the
CPU), so f(x, u) agrees to a few ulp, not bit for bit.  Finite differences amplify that:
  first order  (eps = 1.5e-8):  |d f_x| <~ ulp(f) / eps                ~ 1e-7
  second order (eps = 1.2e-4):  |d f_xx| <~ 2 ulp(f) / eps^2 + 2 |d f_x| / eps ~ 1e-8 + 2e-3 * (f_x noise / 1e-7)
which is why f_x / f_u are held to 2e-6 and the tensors are compared (a) loosely end to end and (b) tightly
(1e-5) with the oracle's own f, f_x, f_u resident, which isolates the second-order stencil.
"""
import numpy as np
import pytest

from problems import initial_trajectory, make, random_state
from synth import rel_err

X_TOL = 1e-10


def _upload_traj(ctx, xs, us, b=0):
    ctx.upload("X", xs, b, 1)
    ctx.upload("U", us, b, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pendulum", "chain6", "tree38", "chain6ff", "tree38ff"])
def test_single_step_dynamics(gpu, name):
    """eval_to (problem.hpp:441-461) at random states: the from-scratch device ABA vs the oracle's, a few ulp"""
    capi = gpu
    B = 16
    model, spec, o = make(name, 1, batch=B)
    rng = np.random.default_rng(0)
    nx, nv = o.nx, model.nv
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        xs = np.stack([np.concatenate([random_state(model, rng), np.zeros(nx)]) for _ in range(B)]); us = 3.0 * rng.normal(size=(B, nv))
        ctx.upload("X", xs); ctx.upload("U", us)
        ctx.rollout()
        got = ctx.download("X")
        for b in range(B):
            ref = o.eval_f(xs[b, :nx], us[b])
            assert np.array_equal(got[b, :nx], xs[b, :nx])
            assert rel_err(got[b, nx:], ref) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,sigma", [("pendulum", 50, 1.0), ("chain6", 100, 0.05), ("tree38", 20, 1.0), ("chain6ff", 30, 0.05),
                                          ("tree38ff", 20, 1.0)])
def test_rollout_parity(gpu, name, T, sigma):
    # long open-loop rollouts of the light UR5-like wrist are chaotic for large torques: keep them gentle,
    # the arithmetic itself is pinned by test_single_step_dynamics
    capi = gpu
    model, spec, o = make(name, T, batch=2)
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        refs = []
        for b in range(2):
            x0, us, xs = initial_trajectory(o, model, seed=10 + b, u_sigma=sigma)
            bad = np.full_like(xs, np.nan); bad[:o.nx] = x0
            _upload_traj(ctx, bad, us, b)
            refs.append(xs)
        ctx.rollout()
        got = ctx.download("X")
        for b in range(2):
            assert rel_err(got[b], refs[b]) < X_TOL


DERIV_SEQS = {"lfx": "LFX", "lfxx": "LFXX", "lx": "LX", "lu": "LU", "lxx": "LXX", "lux": "LUX", "luu": "LUU",
              "f_val": "F_VAL", "fx": "FX", "fu": "FU", "eq_val": "EQ_VAL", "eq_x": "EQ_X", "eq_u": "EQ_U"}
TENSOR_SEQS = {"fxx": "FXX", "fux": "FUX", "fuu": "FUU", "eq_xx": "EQ_XX", "eq_ux": "EQ_UX", "eq_uu": "EQ_UU"}


def _abs_err(ctx, d, key, seq):
    sz = ctx.seq_size(seq)
    if sz == 0:
        return 0.0, 0.0
    got = ctx.download(seq, 0, 1)[0]
    return float(np.max(np.abs(got - d[key][:sz]))), float(np.max(np.abs(d[key][:sz])))


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,fd_mode", [
    ("pendulum", 50, 2), ("pendulum", 50, 1), ("pendulum", 9, 0),
    ("chain6", 10, 2), ("chain6_frame", 10, 2), ("tree38", 4, 2), ("tree38", 5, 0),
    ("tree38_frame", 4, 2), ("tree38_frame", 5, 0), ("tree38_config", 3, 2),
    # Lie-group configurations (free-flyer root): integrate_x / difference_out on SE(3) in every stencil
    ("chain6ff", 4, 2), ("chain6ff_frame", 5, 2), ("tree38ff", 2, 2), ("tree38ff", 3, 0), ("tree38ff_frame", 4, 2),
])
def test_linearize_parity(gpu, name, T, fd_mode):
    _linearize_parity(gpu, name, T, fd_mode, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("model_seed", [7, 12345])
def test_linearize_parity_other_tree_seeds(gpu, model_seed):
    """the compiled-in topology fixes the tree, not its numbers: other seeded masses / placements / inertias"""
    # (the bound is stated in ulps of f; on some seeded models the intermediates of the ABA are larger than f, and the
    # device and the oracle -- run-time-tree kernels included -- sit 9-10 ulp apart instead of < 8)
    _linearize_parity(gpu, "tree38", 3, 2, model_seed, ulps=16)


def _linearize_parity(gpu, name, T, fd_mode, model_seed, ulps=8):
    capi = gpu
    model, spec, o = make(name, T, fd_mode=fd_mode, seed=model_seed)
    x0, us, xs = initial_trajectory(o, model, seed=3, u_sigma=0.05 if name.startswith("chain6") else 0.5)
    d = o.compute_derivatives(xs, us)
    # FD noise bounds (module docstring), scaled by the magnitude of f
    EPS, E1, E2 = 2.220446049250313e-16, 1.4901161193847656e-08, 1.220703125e-04
    fscale = max(1.0, float(np.max(np.abs(d["f_val"]))))
    tol_first = ulps * EPS * fscale / E1
    tol_second_iso = 8 * ulps * EPS * fscale / (E2 * E2)
    tol_second_e2e = tol_second_iso + 4 * tol_first / E2
    with capi.Context(spec) as ctx:
        _upload_traj(ctx, xs, us)
        ctx.linearize()
        analytic = model.kind == capi.MODEL_PENDULUM
        for key, seq in DERIV_SEQS.items():
            err, scale = _abs_err(ctx, d, key, seq)
            if key in ("lfx", "lfxx", "lx", "lu", "lxx", "lux", "luu"):
                assert err == 0.0, key                       # plain arithmetic on the inputs: bit exact
            elif key in ("f_val", "eq_val"):
                assert err <= 1e-12 * max(scale, 1.0), (key, err)
            else:
                tol = 1e-12 if analytic and key in ("fx", "fu") else tol_first * (4 if key.startswith("eq") else 1)
                assert err <= tol * max(scale, 1.0), (key, err, scale)
        # end to end, every stage on the GPU: bounded by the first-order noise / eps (see module docstring)
        for key, seq in TENSOR_SEQS.items():
            err, scale = _abs_err(ctx, d, key, seq)
            tol = tol_second_iso if analytic else tol_second_e2e
            if key.startswith("eq"):
                tol *= 8      # two chained dynamics steps per constraint evaluation
            if fd_mode == 0:
                assert err == 0.0 and scale == 0.0
            else:
                assert err <= tol * max(scale, 1.0), (key, err, scale, tol)
        if fd_mode == 0:
            return
        # second-order stencil in isolation: oracle's f, f_x, f_u (and eq first order) resident
        for key in ("f_val", "fx", "fu", "eq_val", "eq_x", "eq_u"):
            sz = ctx.seq_size(DERIV_SEQS[key])
            if sz:
                ctx.upload(DERIV_SEQS[key], d[key][:sz], 0, 1)
        for seq in TENSOR_SEQS.values():
            if ctx.seq_size(seq):
                ctx.fill(seq, np.nan)
        ctx.linearize(capi.LIN_SECOND)
        for key in ("fxx", "fux", "fuu"):
            err, scale = _abs_err(ctx, d, key, TENSOR_SEQS[key])
            assert err <= tol_second_iso * max(scale, 1.0), (key, err, scale, tol_second_iso)


@pytest.mark.gpu
@pytest.mark.parametrize("name,T", [("pendulum", 50), ("chain6", 10), ("chain6_frame", 12), ("tree38", 12),
                                    ("tree38_frame", 12), ("tree38_config", 6), ("chain6ff_frame", 8), ("tree38ff_frame", 8)])
def test_cost_seq_aug_parity(gpu, name, T):
    capi = gpu
    model, spec, o = make(name, T)
    x0, us, xs = initial_trajectory(o, model, seed=4, u_sigma=0.5)
    rng = np.random.default_rng(0)
    mults = o.alloc_affine(o.Etot)
    mults["origin"][:] = _jitter_states(o, model, xs[:T * o.nx], rng)
    mults["val"][:o.Etot] = rng.normal(size=o.Etot)
    mults["jac"][:o.Etot * o.n] = rng.normal(size=o.Etot * o.n)
    ref = o.cost_seq_aug(xs, us, mults, mu=37.0)
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        _upload_traj(ctx, xs, us)
        for k, s in (("origin", "MULT_ORIGIN"), ("val", "MULT_VAL"), ("jac", "MULT_JAC")):
            if ctx.seq_size(s):
                ctx.upload(s, mults[k][:ctx.seq_size(s)], 0, 1)
        ctx.cost_seq_aug(0, 37.0)
        got = ctx.download("COSTS_OLD", 0, 1)[0]
        assert rel_err(got, ref) < 1e-11
        assert got[T] == 0.0


def _jitter_states(o, model, xs, rng, scale=0.01):
    """xs (+) small random tangent steps: a nearby sequence of valid states (quaternions stay unit)"""
    nq, nv = getattr(model, "nq", model.nv), model.nv
    out = np.array(xs, dtype=float).copy()
    for t in range(out.size // (nq + nv)):
        x = out[t * (nq + nv):(t + 1) * (nq + nv)]
        x[:nq] = o.integrate(x[:nq], scale * rng.normal(size=nv))
        x[nq:] += scale * rng.normal(size=nv)
    return out


def _one_iteration_inputs(o, model, seed, mu, u_sigma, jac_sigma):
    """derivatives + a backward sweep from the oracle: inputs of the forward pass"""
    x0, us, xs = initial_trajectory(o, model, seed=seed, u_sigma=u_sigma)
    d = o.compute_derivatives(xs, us)
    rng = np.random.default_rng(seed)
    mults = o.alloc_affine(o.Etot)
    mults["origin"][:] = xs[:o.T * o.nx]
    mults["jac"][:o.Etot * o.n] = jac_sigma * rng.normal(size=o.Etot * o.n)
    bw = o.backward(d, xs, mults, reg=0.0, mu=mu)
    assert bw["restarts"] == 0, bw["restarts"]      # a well-posed sweep: the gains are meaningful
    return xs, us, d, mults, bw


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,fd_mode,mu,u_sigma,jac_sigma,k_scale", [
    ("pendulum", 50, 2, 100.0, 0.3, 0.1, 1.0),
    ("chain6", 10, 2, 100.0, 0.05, 0.01, 1.0),
    ("chain6", 10, 2, 100.0, 0.05, 0.01, 40.0),     # overshooting feed-forward: the line search has to halve
    ("tree38", 10, 0, 1.0, 0.3, 0.0, 1.0),
    ("tree38", 10, 0, 1.0, 0.3, 0.0, 30.0),
    ("tree38_frame", 10, 0, 100.0, 0.3, 0.01, 1.0),
    ("tree38_frame", 10, 0, 100.0, 0.3, 0.01, 30.0),
    ("chain6ff", 10, 0, 1.0, 0.05, 0.0, 1.0),           # x_new (-) x_old on SE(3) (ddp_fwd.ipp:45 with model_t::difference)
    ("tree38ff", 10, 0, 1.0, 0.3, 0.0, 1.0),
    ("tree38ff", 10, 0, 1.0, 0.3, 0.0, 30.0),
    ("tree38ff_frame", 10, 0, 100.0, 0.3, 0.01, 1.0),
])
def test_forward_parity(gpu, name, T, fd_mode, mu, u_sigma, jac_sigma, k_scale):
    """Same accepted step as the reference's sequential halving, same new trajectory (ddp_fwd.ipp:9-67)."""
    capi = gpu
    model, spec, o = make(name, T, fd_mode=fd_mode)
    xs, us, d, mults, bw = _one_iteration_inputs(o, model, 21, mu, u_sigma, jac_sigma)
    bw["fb"]["val"] *= k_scale
    step_ref, xs_ref, us_ref, n_evals = o.forward(xs, us, mults, bw["fb"], bw["mu"])
    if k_scale > 1:
        assert n_evals > 1 and step_ref < 1.0, (step_ref, n_evals)    # the case really exercises the halving
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        _upload_traj(ctx, xs, us)
        ctx.upload("X_NEW", xs, 0, 1)
        ctx.upload("U_NEW", us, 0, 1)
        for k, s in (("origin", "MULT_ORIGIN"), ("val", "MULT_VAL"), ("jac", "MULT_JAC")):
            if ctx.seq_size(s):
                ctx.upload(s, mults[k][:ctx.seq_size(s)], 0, 1)
        for k, s in (("origin", "FB_ORIGIN"), ("val", "FB_VAL"), ("jac", "FB_JAC")):
            ctx.upload(s, bw["fb"][k][:ctx.seq_size(s)], 0, 1)
        rc, step, dcost = ctx.forward(bw["mu"], n_alpha=8)
        assert step[0] == step_ref, (step, step_ref, n_evals)
        assert rel_err(ctx.download("X_NEW", 0, 1)[0], xs_ref) < 1e-9
        assert rel_err(ctx.download("U_NEW", 0, 1)[0], us_ref) < 1e-9
        dc_ref, _, _ = o.forward_alpha(step_ref, xs, us, mults, bw["fb"], bw["mu"])
        assert dcost[0] <= 0 and abs(dcost[0] - dc_ref) <= 1e-9 * max(1.0, abs(dc_ref))


@pytest.mark.gpu
@pytest.mark.parametrize("n_alpha", [1, 3, 4, 5, 8])
def test_forward_any_number_of_batched_steps(gpu, n_alpha):
    """The accept decision is the sequential halving's whatever the number of step sizes rolled out together (ddp_fwd.ipp:29-64):
    rounds of n_alpha candidates.  On the latency kernel this also exercises workgroups with no live candidate (n_alpha <= 4:
    the second workgroup of an instance leaves at once) and partly filled ones."""
    capi = gpu
    T = 10
    model, spec, o = make("tree38", T, fd_mode=0)
    xs, us, d, mults, bw = _one_iteration_inputs(o, model, 21, 1.0, 0.3, 0.0)
    bw["fb"]["val"] *= 30.0                                      # overshooting feed-forward: several halvings
    step_ref, xs_ref, us_ref, n_evals = o.forward(xs, us, mults, bw["fb"], bw["mu"])
    assert n_evals > 1
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        assert ctx.info()["fwd_path"] == 1
        _upload_traj(ctx, xs, us)
        ctx.upload("X_NEW", xs, 0, 1)
        ctx.upload("U_NEW", us, 0, 1)
        for k, s_ in (("origin", "FB_ORIGIN"), ("val", "FB_VAL"), ("jac", "FB_JAC")):
            ctx.upload(s_, bw["fb"][k][:ctx.seq_size(s_)], 0, 1)
        rc, step, dcost = ctx.forward(bw["mu"], n_alpha=n_alpha)
        assert step[0] == step_ref, (n_alpha, step, step_ref)
        assert rel_err(ctx.download("X_NEW", 0, 1)[0], xs_ref) < 1e-9
        assert rel_err(ctx.download("U_NEW", 0, 1)[0], us_ref) < 1e-9


@pytest.mark.gpu
def test_forward_line_search_floor(gpu):
    """A feedback that can only increase the cost: every candidate down to 2^-33 is rejected, the call reports
    the floor, returns step = 2^-34 and leaves the last tried rollout in X_NEW (ddp_fwd.ipp:35-37,59)."""
    capi = gpu
    T = 6
    model, spec, o = make("pendulum", T, fd_mode=0)
    spec.eq_kind = capi.EQ_NONE; spec.ne[:] = 0; spec.Etot = 0
    us = np.zeros(T)
    xs = o.rollout(np.zeros(2), us)
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        _upload_traj(ctx, xs, us)
        ctx.upload("X_NEW", xs, 0, 1); ctx.upload("U_NEW", us, 0, 1)
        ctx.upload("FB_ORIGIN", xs[:T * 2], 0, 1)
        ctx.upload("FB_VAL", np.ones(T), 0, 1)       # u = 0 is optimal for l = c/2 u^2: any step costs more
        ctx.upload("FB_JAC", np.zeros(T * 2), 0, 1)
        rc, step, dcost = ctx.forward(1.0, n_alpha=8)
        assert rc == capi.EV_LINESEARCH_FLOOR
        assert step[0] == 2.0 ** -34
        assert np.allclose(ctx.download("U_NEW", 0, 1)[0], 2.0 ** -33)


@pytest.mark.gpu
def test_static_vs_generic_linearisation_chain6(gpu, monkeypatch):
    """The second compiled-in topology (UR5-like chain, BASELINE config 2 horizon T = 100): static vs run-time-tree
    stencil kernels on the same trajectories, constraint stage included."""
    capi = gpu
    T, B = 100, 4
    model, spec, o = make("chain6", T, batch=B, fd_mode=2)
    rng = np.random.default_rng(9)
    us = 0.05 * rng.normal(size=(B, T * model.nv))
    out = {}
    seqs = ("F_VAL", "FX", "FU", "FXX", "FUX", "FUU", "EQ_VAL", "EQ_X", "EQ_U", "EQ_XX")
    for tag in ("static", "generic"):
        if tag == "generic":
            monkeypatch.setenv("DDP_HIP_NO_STATIC", "1")
        with capi.Context(spec) as ctx:
            ctx.upload("X", np.zeros((B, (T + 1) * 2 * model.nv)))
            ctx.upload("U", us)
            ctx.rollout()
            ctx.linearize()
            out[tag] = {s: ctx.download(s) for s in seqs}
    monkeypatch.delenv("DDP_HIP_NO_STATIC")
    EPS, E1, E2 = 2.220446049250313e-16, 1.4901161193847656e-08, 1.220703125e-04
    fscale = max(1.0, float(np.max(np.abs(out["generic"]["F_VAL"]))))
    assert np.array_equal(out["static"]["F_VAL"], out["generic"]["F_VAL"])
    tol1 = 32 * EPS * fscale / E1
    tol2 = 64 * EPS * fscale / (E2 * E2) + 4 * tol1 / E2
    for s in seqs[1:]:
        a, g = out["static"][s], out["generic"][s]
        tol = tol1 if s in ("FX", "FU", "EQ_VAL", "EQ_X", "EQ_U") else tol2
        if s.startswith("EQ"):
            tol *= 8
        assert np.all(np.isfinite(a)), s
        assert float(np.max(np.abs(a - g))) <= tol * max(1.0, float(np.max(np.abs(g)))), (s, float(np.max(np.abs(a - g))), tol)


@pytest.mark.gpu
def test_static_vs_generic_linearisation_full_horizon(gpu, monkeypatch):
    """The static-topology stencil kernels (lin_static.hip) against the generic run-time-tree kernels (lin.hip) at the
    full horizon, with more (instance, t) pairs than one workspace slice holds (1 024): same stencil, same caches, the
    same operation sequences up to the order of a few additions -- held to a few ulp of f amplified by 1 / eps^2."""
    capi = gpu
    T, B = 200, 6
    model, spec, o = make("tree38", T, batch=B, fd_mode=2)
    rng = np.random.default_rng(5)
    us = 0.3 * rng.normal(size=(B, T * model.nv))
    out = {}
    for tag in ("static", "generic"):
        if tag == "generic":
            monkeypatch.setenv("DDP_HIP_NO_STATIC", "1")
        with capi.Context(spec) as ctx:
            ctx.upload("X", np.zeros((B, (T + 1) * 2 * model.nv)))
            ctx.upload("U", us)
            ctx.rollout()
            ctx.linearize()
            out[tag] = {k: ctx.download(s) for k, s in (("f_val", "F_VAL"), ("fx", "FX"), ("fu", "FU"))}
            # the tensors are 1.2 GB per instance: compare instance by instance
            for b in (0, B - 1):
                for k, s in (("fxx", "FXX"), ("fux", "FUX"), ("fuu", "FUU")):
                    out[tag][k, b] = ctx.download(s, b, 1)[0]
    monkeypatch.delenv("DDP_HIP_NO_STATIC")
    EPS, E1, E2 = 2.220446049250313e-16, 1.4901161193847656e-08, 1.220703125e-04
    fscale = max(1.0, float(np.max(np.abs(out["generic"]["f_val"]))))
    assert np.array_equal(out["static"]["f_val"], out["generic"]["f_val"])
    for k in ("fx", "fu"):
        err = float(np.max(np.abs(out["static"][k] - out["generic"][k])))
        assert err <= 32 * EPS * fscale / E1, (k, err)      # FMA contraction differs with the code shape: a few ulp of f
    for b in (0, B - 1):
        for k in ("fxx", "fux", "fuu"):
            a, g = out["static"][k, b], out["generic"][k, b]
            assert np.all(np.isfinite(a))
            err = float(np.max(np.abs(a - g)))
            tol = 64 * EPS * fscale / (E2 * E2) + 4 * (32 * EPS * fscale / E1) / E2
            assert err <= tol * max(1.0, float(np.max(np.abs(g)))), (k, b, err, tol)
        # size-independent properties of the stencil: a pair (i, j) is evaluated once and written to both (j, k) orders
        # (problem.hpp:292-295), so f_xx and f_uu are exactly symmetric in their last two indices at every t
        n, m = 2 * model.nv, model.nv
        fxx = out["static"]["fxx", b].reshape(T, n, n, n)       # [t][k][j][i] for element (i, j, k) at i + j n + k n n
        assert np.array_equal(fxx, fxx.transpose(0, 2, 1, 3))
        fuu = out["static"]["fuu", b].reshape(T, m, m, n)
        assert np.array_equal(fuu, fuu.transpose(0, 2, 1, 3))
        # q+ = q + dt v is linear in x and does not see u: its rows of every tensor carry rounding noise only
        assert float(np.max(np.abs(out["static"]["fux", b].reshape(T, n, m, n)[..., :model.nv]))) <= tol
