"""Outer augmented-Lagrangian loop pieces on the device (csrc/outer.hip) vs the CPU oracle:
update_origin (mat_seq_common.hpp:62-89), optimality_obj / optimality_constr (ddp.hpp:576-627, 516-523) and the
multiplier update of update_derivatives (ddp.hpp:680-688).  Plain double arithmetic on identical inputs, sums taken in
the same order: held to 1e-12 relative (north star: 1e-10 on V_x / K)."""
import numpy as np
import pytest

from problems import initial_trajectory, make
from test_dynamics_parity import _jitter_states
from synth import rel_err

DERIV_UP = {"lfx": "LFX", "lx": "LX", "lu": "LU", "fx": "FX", "fu": "FU", "eq_val": "EQ_VAL", "eq_x": "EQ_X", "eq_u": "EQ_U"}


def _setup(capi, name, T, B, seed):
    model, spec, o = make(name, T, batch=B, fd_mode=0)
    rng = np.random.default_rng(seed)
    cases = []
    for b in range(B):
        x0, us, xs = initial_trajectory(o, model, seed=seed + b, u_sigma=0.05 if name.startswith("chain6") else 0.3)
        d = o.compute_derivatives(xs, us)
        # a non-trivial cost gradient and terminal gradient (the reference's cost has l_x = 0)
        d["lx"][:T * o.n] = 0.1 * rng.normal(size=T * o.n)
        d["lfx"][:o.n] = 0.2 * rng.normal(size=o.n)
        mults = o.alloc_affine(o.Etot)
        mults["origin"][:] = _jitter_states(o, model, xs[:T * o.nx], rng)
        mults["val"][:o.Etot] = rng.normal(size=o.Etot)
        mults["jac"][:o.Etot * o.n] = 0.3 * rng.normal(size=o.Etot * o.n)
        fb = o.alloc_affine(T * o.m)
        fb["origin"][:] = _jitter_states(o, model, xs[:T * o.nx], rng)
        fb["val"][:] = 0.1 * rng.normal(size=T * o.m)
        fb["jac"][:] = 0.1 * rng.normal(size=T * o.m * o.n)
        cases.append((xs, us, d, mults, fb))
    return model, spec, o, cases


def _upload(ctx, b, xs, us, d, mults, fb):
    ctx.upload("X", xs, b, 1)
    ctx.upload("U", us, b, 1)
    for k, s in DERIV_UP.items():
        sz = ctx.seq_size(s)
        if sz:
            ctx.upload(s, d[k][:sz], b, 1)
    for pre, a in (("MULT", mults), ("FB", fb)):
        for k in ("origin", "val", "jac"):
            s = f"{pre}_{k.upper()}"
            sz = ctx.seq_size(s)
            if sz:
                ctx.upload(s, a[k][:sz], b, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,T", [("pendulum", 50), ("chain6", 10), ("chain6_frame", 12), ("tree38_frame", 8), ("tree38_config", 4),
                                    ("tree38", 6), ("chain6ff_frame", 8), ("tree38ff_frame", 6)])
def test_outer_loop_pieces(gpu, name, T):
    capi = gpu
    B, mu = 2, 37.0
    model, spec, o, cases = _setup(capi, name, T, B, seed=11)
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        for b, c in enumerate(cases):
            _upload(ctx, b, *c)
        # optimality measures
        obj, constr = ctx.optimality(mu)
        for b, (xs, us, d, mults, fb) in enumerate(cases):
            o_ref, c_ref = o.optimality(xs, mults, mu, d)
            assert abs(obj[b] - o_ref) <= 1e-12 * max(1.0, abs(o_ref)), (b, obj[b], o_ref)
            assert abs(constr[b] - c_ref) <= 1e-12 * max(1.0, abs(c_ref)), (b, constr[b], c_ref)
        # multiplier update with the resident feedback (ddp.hpp:680-688), against a numpy restatement of the same lines
        ctx.update_multipliers(mu)
        for b, (xs, us, d, mults, fb) in enumerate(cases):
            val, jac = mults["val"].copy(), mults["jac"].copy()
            E = 0
            for t in range(T):
                e = int(o.ne[t])
                if e:
                    eqv = d["eq_val"][E:E + e]
                    eqx = d["eq_x"][E * o.n:(E + e) * o.n].reshape(e, o.n, order="F")
                    equ = d["eq_u"][E * o.m:(E + e) * o.m].reshape(e, o.m, order="F")
                    k = fb["val"][t * o.m:(t + 1) * o.m]
                    K = fb["jac"][t * o.m * o.n:(t + 1) * o.m * o.n].reshape(o.m, o.n, order="F")
                    val[E:E + e] += mu * (eqv + equ @ k)
                    jac[E * o.n:(E + e) * o.n] += (mu * (eqx + equ @ K)).ravel(order="F")
                E += e
            if o.Etot:
                assert rel_err(ctx.download("MULT_VAL", b, 1)[0], val[:o.Etot]) < 1e-12
                assert rel_err(ctx.download("MULT_JAC", b, 1)[0], jac[:o.Etot * o.n]) < 1e-12
                ctx.upload("MULT_VAL", mults["val"][:o.Etot], b, 1)          # restore for the next check
                ctx.upload("MULT_JAC", mults["jac"][:o.Etot * o.n], b, 1)
        # update_origin of both affine sequences at x_new = X.  Vector-space states leave the jacobians untouched; with a
        # free-flyer root they are multiplied by d difference(origin, .)/d finish at the old origin (mat_seq_common.hpp:77-84)
        lie = o.nx != o.n
        ctx.update_origin(0)
        ctx.update_origin(1)
        for b, (xs, us, d, mults, fb) in enumerate(cases):
            m_ref = o.update_origin(mults, o.ne, xs)
            f_ref = o.update_origin(fb, np.full(T, o.m, dtype=np.int64), xs)
            assert np.array_equal(ctx.download("MULT_ORIGIN", b, 1)[0], xs[:T * o.nx])
            assert np.array_equal(ctx.download("FB_ORIGIN", b, 1)[0], xs[:T * o.nx])
            if o.Etot:
                assert rel_err(ctx.download("MULT_VAL", b, 1)[0], m_ref["val"][:o.Etot]) < 1e-12
                if lie:
                    assert rel_err(ctx.download("MULT_JAC", b, 1)[0], m_ref["jac"][:o.Etot * o.n]) < 1e-12
                else:
                    assert np.array_equal(ctx.download("MULT_JAC", b, 1)[0], mults["jac"][:o.Etot * o.n])
            assert rel_err(ctx.download("FB_VAL", b, 1)[0], f_ref["val"][:T * o.m]) < 1e-12
            if lie:
                assert rel_err(ctx.download("FB_JAC", b, 1)[0], f_ref["jac"][:T * o.m * o.n]) < 1e-12
                assert not np.array_equal(f_ref["jac"][:T * o.m * o.n], fb["jac"][:T * o.m * o.n])
            else:
                assert np.array_equal(ctx.download("FB_JAC", b, 1)[0], fb["jac"][:T * o.m * o.n])


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,iters,mu,w,n,tol", [
    ("pendulum", 50, 12, 10.0, 1.0, 10.0, 1e-6),      # test/pendulum_ddp.cpp shape (double-appropriate parameters)
    ("chain6", 10, 6, 1e4, 1e-1, 10.0, 1e-4),         # test/pinocchio_ddp.cpp shape: FD first order on both sides
])
def test_whole_solve_against_oracle(gpu, name, T, iters, mu, w, n, tol):
    """solve<M> (ddp.hpp:745-842) end to end: every sequence operation on the device (ddp_pinocchio_amd/solver.py) vs
    the oracle's restatement of the same loop, same initial multipliers.  The loop amplifies the finite-difference
    noise of the derivatives from iteration to iteration, hence the looser tolerance on the final trajectory."""
    from ddp_pinocchio_amd import solver
    capi = gpu
    B = 2
    model, spec, o = make(name, T, batch=B, fd_mode=2)
    x0 = np.zeros(2 * model.nv)
    us0 = np.zeros(T * model.nv)
    xs0 = o.rollout(x0, us0)
    rng = np.random.default_rng(3)
    seed = 0.01 * rng.normal(size=max(o.Etot * o.n, 1))
    xs_ref, us_ref, fb_ref, log_ref = o.solve(xs0, us0, seed, max_iterations=iters, threshold=1e-8, mu=mu, reg=0.0, w=w, n=n)
    with capi.Context(spec) as ctx:
        for b in range(B):
            ctx.upload("X", xs0, b, 1); ctx.upload("U", us0, b, 1)
            ctx.upload("X_NEW", xs0, b, 1); ctx.upload("U_NEW", us0, b, 1)
            ctx.upload("MULT_ORIGIN", xs0[:T * o.nx], b, 1)
            ctx.upload("MULT_VAL", np.zeros(o.Etot), b, 1)
            ctx.upload("MULT_JAC", seed[:o.Etot * o.n], b, 1)
        log = solver.solve(ctx, iters, 1e-8, mu, 0.0, w, n)
        xs = ctx.download("X")
        us = ctx.download("U")
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(us[0], us[1])         # instances are independent and deterministic
    assert log["mu"][0] == log_ref["mu"], (log["mu"], log_ref["mu"])
    assert rel_err(xs[0], xs_ref) < tol, rel_err(xs[0], xs_ref)
    assert rel_err(us[0], us_ref) < 10 * tol, rel_err(us[0], us_ref)
    assert abs(log["opt_constr"][0] - log_ref["opt_constr"]) <= 100 * tol * max(1.0, log_ref["opt_constr"])
