"""Free-flyer (SE(3)) root joint -- SURVEY.md 8f-2: the Lie-group layer of the reference's model wrapper
(pinocchio_model.ipp:244-340: integrate, difference, d_integrate_dq/dv, d_difference_dq_start/finish) and every place the hot
path goes through it: eval_f (problem.hpp:414-460), the finite-difference stencils (problem.hpp:67-150, 463-503:
integrate_x / difference_out), the forward pass' x_new (-) x_old (ddp_fwd.ipp:45) and update_origin
(mat_seq_common.hpp:62-89).

CPU part (oracle pinning).  The Lie operations live in the absent pinocchio; the pins are
  (i)   the reference's own property tests, test/pinocchio.cpp:17-57 ("integrate-difference") and :59-100 ("affine-function"),
        run on the oracle with the reference's tolerances (eps * 1e3; the mpfr step 1e-50 becomes 1e-5 in double);
  (ii)  an independent route: homogeneous 4x4 matrices with mpmath's generic matrix exponential / logarithm (no closed forms);
  (iii) the tangent-space jacobians against central differences of (i)-(ii)-checked maps;
  (iv)  closed forms of the dynamics with a free root (free fall; RNEA o ABA = id).
No reference output exists for a free-flyer run in double (the reference's free-flyer cases run in mpfr): beyond these pins
the free-flyer path is "parity unpinned" against the reference itself -- stated in DESIGN.md.

GPU part: the device path against the oracle on the free-flyer builtins (the remaining free-flyer parity cases are the
chain6ff / tree38ff rows of test_dynamics_parity.py and test_outer_parity.py)."""
import numpy as np
import pytest

from problems import initial_trajectory, make, neutral_state, random_state
from synth import rel_err

EPS = 2.220446049250313e-16


def _oracle(which, T=1):
    from ddp_pinocchio_amd import capi
    from oracle.binding import Oracle
    bm = capi.BuiltinModel(which, 1)
    return bm, Oracle(bm, T)


def _is_approx(a, b, prec):
    """Eigen's isApprox: |a - b| <= prec * min(|a|, |b|)"""
    return np.linalg.norm(a - b) <= prec * min(np.linalg.norm(a), np.linalg.norm(b))


FF_MODELS = [3, 4]      # BUILTIN_CHAIN6_FF, BUILTIN_TREE38_FF


@pytest.mark.parametrize("which", FF_MODELS)
def test_integrate_difference_round_trip(which):
    """test/pinocchio.cpp:17-57"""
    bm, o = _oracle(which)
    assert bm.ff and bm.nq == bm.nv + 1 and o.nx == bm.nq + bm.nv
    rng = np.random.default_rng(5)
    for _ in range(20):
        q0 = random_state(bm, rng)[:bm.nq]
        v0 = rng.uniform(-1, 1, size=bm.nv)
        q1 = o.integrate(q0, v0)
        assert abs(np.linalg.norm(q1[3:7]) - 1.0) < 4 * EPS
        v1 = o.difference(q0, q1)
        assert _is_approx(v0, v1, EPS * 1e3)
        q2 = o.integrate(q1, -v0)
        # q and -q are the same rotation: compare up to the sign of the quaternion, as pinocchio's isSameConfiguration does
        if np.dot(q2[3:7], q0[3:7]) < 0:
            q2[3:7] = -q2[3:7]
        assert _is_approx(q2, q0, EPS * 1e3)


@pytest.mark.parametrize("which", FF_MODELS)
def test_affine_function_first_order_consistency(which):
    """test/pinocchio.cpp:59-100: difference(q0, q1 (+) dv) = difference(q0, q1) + J dv + O(dh^2), J = d_difference_dq_finish"""
    bm, o = _oracle(which)
    rng = np.random.default_rng(6)
    dh = 1e-5
    for _ in range(10):
        q0, q1 = random_state(bm, rng)[:bm.nq], random_state(bm, rng)[:bm.nq]
        v1 = o.difference(q0, q1)
        J = o.d_difference_dq_finish(q0, q1)
        dv = rng.uniform(-1, 1, size=bm.nv) * dh
        v2 = o.difference(q0, o.integrate(q1, dv))
        f0, j0 = rng.uniform(-1, 1), rng.uniform(-1, 1, size=bm.nv)
        assert np.linalg.norm(v2 - v1 - J @ dv) < dh * dh * 1e2
        assert abs((f0 + j0 @ v2) - (f0 + j0 @ v1 + (j0 @ J) @ dv)) < dh * dh * 1e2


def _hom(q):
    """q = [p, quat xyzw] -> 4x4 (pinocchio's free-flyer configuration layout)"""
    import mpmath as mp
    x, y, z, w = [mp.mpf(float(c)) for c in q[3:7]]
    R = mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = mp.eye(4)
    for i in range(3):
        for j in range(3):
            M[i, j] = R[i, j]
        M[i, 3] = mp.mpf(float(q[i]))
    return M


def _hat(v):
    import mpmath as mp
    lx, ly, lz, wx, wy, wz = [mp.mpf(float(c)) for c in v]
    return mp.matrix([[0, -wz, wy, lx], [wz, 0, -wx, ly], [-wy, wx, 0, lz], [0, 0, 0, 0]])


def test_se3_against_matrix_exponential():
    """integrate = q * exp6(v), difference = log6(q0^-1 q1), v = [linear, angular] in the body frame: against mpmath's generic
    expm / logm on 4x4 homogeneous matrices at 40 digits (no Rodrigues formula, no quaternions on that side)"""
    import mpmath as mp
    mp.mp.dps = 40
    bm, o = _oracle(3)
    rng = np.random.default_rng(8)
    for k in range(8):
        q = random_state(bm, rng)[:bm.nq]
        v = rng.normal(size=bm.nv) * (1e-6 if k == 0 else 1.0)       # k = 0: the small-angle branch
        q1 = o.integrate(q, v)
        M1 = _hom(q) * mp.expm(_hat(v[:6]), method="pade")
        got = _hom(q1)
        err = max(abs(got[i, j] - M1[i, j]) for i in range(4) for j in range(4))
        assert err < 1e-14, (k, err)
        assert np.array_equal(q1[7:], q[7:] + v[6:])                  # the revolute joints behind the root: plain addition
        # difference against logm
        q0 = random_state(bm, rng)[:bm.nq]
        d = o.difference(q0, q1)
        L = mp.logm(mp.inverse(_hom(q0)) * _hom(q1))
        ref = np.array([float(mp.re(L[0, 3])), float(mp.re(L[1, 3])), float(mp.re(L[2, 3])),
                        float(mp.re(L[2, 1])), float(mp.re(L[0, 2])), float(mp.re(L[1, 0]))])
        assert np.max(np.abs(d[:6] - ref)) < 1e-12, (k, np.max(np.abs(d[:6] - ref)))
        assert np.allclose(d[6:], q1[7:] - q0[7:], rtol=0, atol=1e-15)


def test_so3_series_coefficients_against_mpmath():
    """The closed forms of (1 - cos t)/t^2, (t - sin t)/t^3, (1 - (t/2) cot(t/2))/t^2, (1 - t^2/2 - cos t)/t^4 and
    (t - sin t - t^3/6)/t^5 cancel near t = 0: round 2 switched from the series to them at t = 1e-4 / 1e-3, where they are only
    good to 2e-8 / 3e-5 (d = 0.337 instead of 0.08333 at t = 1.0001e-4).  Rotation increments of 1e-4 .. 1e-2 rad are what
    se3_difference (forward pass, optimality) and Jlog6 (update_origin) see near convergence.  Swept over t = 1e-6 .. pi
    against mpmath at 50 digits: every coefficient to 5e-13 relative (device code = the same lines, csrc/lie.h)."""
    import ctypes as C
    import mpmath as mp
    from oracle.binding import lib
    mp.mp.dps = 50
    L = lib()
    out = (C.c_double * 6)()
    ts = np.concatenate([np.logspace(-6, np.log10(3.1), 140), [1.0001e-4, 1e-3, 3e-3, 0.2 - 1e-9, 0.2 + 1e-9, 1 - 1e-9, 1 + 1e-9]])
    worst = np.zeros(6)
    for t_ in ts:
        t = mp.mpf(float(t_))
        t2 = float(t_) * float(t_)
        tt = mp.sqrt(mp.mpf(t2))                              # the functions are evaluated at the t^2 the code receives
        ref = [mp.sin(tt) / tt, (1 - mp.cos(tt)) / tt**2, (tt - mp.sin(tt)) / tt**3, (1 - (tt / 2) * mp.cot(tt / 2)) / tt**2,
               (1 - tt**2 / 2 - mp.cos(tt)) / tt**4, (tt - mp.sin(tt) - tt**3 / 6) / tt**5]
        L.orc_so3_coeffs(t2, out)
        for k in range(6):
            worst[k] = max(worst[k], abs(float((mp.mpf(out[k]) - ref[k]) / ref[k])))
    assert np.all(worst < 5e-13), worst


def test_se3_difference_and_jlog_at_small_rotations():
    """difference and d_difference_dq_finish (Jlog6) for rotation increments from 1e-6 to 1e-1 rad, against mpmath's generic
    logm at 50 digits and central differences of it -- the range where round 2's thresholds lost eight digits"""
    import mpmath as mp
    mp.mp.dps = 50
    bm, o = _oracle(3)
    rng = np.random.default_rng(11)
    for ang in (1e-6, 1.0001e-4, 1e-3, 3e-3, 1e-2, 1e-1):
        q0 = random_state(bm, rng)[:bm.nq]
        v = np.zeros(bm.nv)
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        v[:3] = rng.normal(size=3) * 0.3
        v[3:6] = ang * axis
        q1 = o.integrate(q0, v)
        d = o.difference(q0, q1)
        Lg = mp.logm(mp.inverse(_hom(q0)) * _hom(q1))
        ref = np.array([float(mp.re(Lg[0, 3])), float(mp.re(Lg[1, 3])), float(mp.re(Lg[2, 3])),
                        float(mp.re(Lg[2, 1])), float(mp.re(Lg[0, 2])), float(mp.re(Lg[1, 0]))])
        # q1 itself carries the rounding of one quaternion product (1e-16 absolute on a unit quaternion)
        assert np.max(np.abs(d[:6] - ref)) < 2e-15 + 1e-13 * np.max(np.abs(ref)), (ang, np.max(np.abs(d[:6] - ref)))
        assert np.max(np.abs(d[:6] - v[:6])) < 5e-15, (ang, np.max(np.abs(d[:6] - v[:6])))
        # Jlog6 against central differences of the mpmath logarithm (right perturbations of q1)
        J = o.d_difference_dq_finish(q0, q1)[:6, :6]
        h = mp.mpf(10) ** -20
        M0i, M1 = mp.inverse(_hom(q0)), _hom(q1)
        for j in range(6):
            e = np.zeros(6); e[j] = 1.0
            E = _hat(e)
            Lp = mp.logm(M0i * M1 * mp.expm(E * h, method="pade"))
            Lm = mp.logm(M0i * M1 * mp.expm(E * (-h), method="pade"))
            col = [(Lp[0, 3] - Lm[0, 3]) / (2 * h), (Lp[1, 3] - Lm[1, 3]) / (2 * h), (Lp[2, 3] - Lm[2, 3]) / (2 * h),
                   (Lp[2, 1] - Lm[2, 1]) / (2 * h), (Lp[0, 2] - Lm[0, 2]) / (2 * h), (Lp[1, 0] - Lm[1, 0]) / (2 * h)]
            col = np.array([float(mp.re(c)) for c in col])
            assert np.max(np.abs(J[:, j] - col)) < 1e-12, (ang, j, np.max(np.abs(J[:, j] - col)))


def test_se3_known_answers():
    """pinocchio's documented conventions: translation is applied in the body frame, quaternion stored x, y, z, w"""
    bm, o = _oracle(3)
    q = o.neutral()
    assert np.array_equal(q[:7], [0, 0, 0, 0, 0, 0, 1])
    v = np.zeros(bm.nv); v[5] = np.pi / 2                             # a quarter turn about z
    q1 = o.integrate(q, v)
    assert np.allclose(q1[3:7], [0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)], atol=1e-16)
    v2 = np.zeros(bm.nv); v2[0] = 1.0                                 # then one unit along the body's x: the world's y
    q2 = o.integrate(q1, v2)
    assert np.allclose(q2[:3], [0, 1, 0], atol=1e-15)


@pytest.mark.parametrize("which", FF_MODELS)
def test_lie_jacobians_against_differences(which):
    """d_integrate_dq / dv and d_difference_dq_start / finish (pinocchio_model.ipp:272-340) in tangent coordinates against
    central differences of integrate / difference: truncation h^2 ~ 1e-10 at h = 1e-5, rounding eps / h ~ 2e-11"""
    bm, o = _oracle(which)
    rng = np.random.default_rng(9)
    h, nv = 1e-5, bm.nv
    E = np.eye(nv)
    q, q1 = random_state(bm, rng)[:bm.nq], random_state(bm, rng)[:bm.nq]
    v = rng.normal(size=nv)
    base = o.integrate(q, v)
    fd_q = np.stack([(o.difference(base, o.integrate(o.integrate(q, h * E[k]), v))
                      - o.difference(base, o.integrate(o.integrate(q, -h * E[k]), v))) / (2 * h) for k in range(nv)], axis=1)
    fd_v = np.stack([(o.difference(base, o.integrate(q, v + h * E[k])) - o.difference(base, o.integrate(q, v - h * E[k]))) / (2 * h)
                     for k in range(nv)], axis=1)
    assert np.max(np.abs(o.d_integrate_dq(q, v) - fd_q)) < 1e-8
    assert np.max(np.abs(o.d_integrate_dv(q, v) - fd_v)) < 1e-8
    fd_s = np.stack([(o.difference(o.integrate(q, h * E[k]), q1) - o.difference(o.integrate(q, -h * E[k]), q1)) / (2 * h)
                     for k in range(nv)], axis=1)
    fd_f = np.stack([(o.difference(q, o.integrate(q1, h * E[k])) - o.difference(q, o.integrate(q1, -h * E[k]))) / (2 * h)
                     for k in range(nv)], axis=1)
    assert np.max(np.abs(o.d_difference_dq_start(q, q1) - fd_s)) < 1e-7
    assert np.max(np.abs(o.d_difference_dq_finish(q, q1) - fd_f)) < 1e-7


@pytest.mark.parametrize("which", FF_MODELS)
def test_free_root_dynamics_identities(which):
    """(a) nothing actuated, at rest: the whole mechanism is in free fall -- the root accelerates with gravity expressed in its
    own frame, no joint moves; (b) RNEA(q, v, ABA(q, v, tau)) = tau with the 6-DoF root"""
    bm, o = _oracle(which)
    rng = np.random.default_rng(10)
    q = random_state(bm, rng)[:bm.nq]
    nv = bm.nv
    a = o.aba(q, np.zeros(nv), np.zeros(nv))
    x, y, z, w = q[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    assert np.max(np.abs(a[:3] - R.T @ np.array([0.0, 0.0, -9.81]))) < 1e-9
    assert np.max(np.abs(a[3:])) < 1e-9
    v, tau = rng.normal(size=nv), 3.0 * rng.normal(size=nv)
    qdd = o.aba(q, v, tau)
    assert rel_err(o.rnea(q, v, qdd), tau) < 1e-10


def test_free_flyer_linearisation_vs_tangent_differences():
    """f_x of the free-flyer model is the jacobian of x -> f(x, u) in tangent coordinates (problem.hpp:463-503 with
    integrate_x / difference_out): the oracle's first-order FD restatement against a wider central difference of its own eval_f
    composed by hand -- guards the layout (q block nq wide in x, nv wide in the tangent)"""
    bm, o = _oracle(3)
    rng = np.random.default_rng(11)
    x, u = random_state(bm, rng, 0.3), rng.normal(size=bm.nv)
    x[3:7] /= np.linalg.norm(x[3:7])
    nq, nv, n = bm.nq, bm.nv, 2 * bm.nv
    fx, fu, f = o.first_order_f(x, u)
    fx = fx.reshape(n, n, order="F")

    def plus(x, dx):
        return np.concatenate([o.integrate(x[:nq], dx[:nv]), x[nq:] + dx[nv:]])

    def minus(a, b):
        return np.concatenate([o.difference(a[:nq], b[:nq]), b[nq:] - a[nq:]])

    f0, h = o.eval_f(x, u), 1e-5
    E = np.eye(n)
    fd = np.stack([(minus(f0, o.eval_f(plus(x, h * E[k]), u)) - minus(f0, o.eval_f(plus(x, -h * E[k]), u))) / (2 * h) for k in range(n)], axis=1)
    # the restatement is the reference's FORWARD difference at eps = 1.5e-8: truncation eps |f''| / 2 ~ 1e-6 on the acceleration rows
    assert np.max(np.abs(fx - fd)) < 1e-5 * max(1.0, np.max(np.abs(fd)))


# ----------------------------------------------------------------------------------------------------------------- device
@pytest.mark.gpu
@pytest.mark.parametrize("which", FF_MODELS)
def test_model_handle_free_flyer(gpu, which):
    """seam B2 point evaluations with a free-flyer root: dynamics_aba and frame_coordinates / d_frame_coordinates"""
    capi = gpu
    from oracle.binding import Oracle
    bm = capi.BuiltinModel(which, 1)
    o = Oracle(bm, 1)
    rng = np.random.default_rng(12)
    q = random_state(bm, rng)[:bm.nq]
    v, tau = rng.normal(size=bm.nv), 2.0 * rng.normal(size=bm.nv)
    with capi.ModelHandle(bm) as h:
        assert rel_err(h.aba(q, v, tau), o.aba(q, v, tau)) < 1e-12
        joint, off = bm.nj - 1, np.array([0.01, -0.02, 0.08])
        p3, J = h.frame(joint, off, q)
        assert rel_err(p3, o.frame_position(joint, off, q)) < 1e-13
        assert float(np.max(np.abs(J - o.frame_jacobian(joint, off, q)))) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name,T", [("chain6ff", 12), ("chain6ff_frame", 12), ("tree38ff", 10), ("tree38ff_frame", 10)])
def test_free_flyer_backward_on_device_derivatives(gpu, name, T):
    """linearise on the device, sweep on the device; the oracle sweeps the SAME (downloaded) derivatives: isolates the Riccati
    recursion with nx != n (the affine origins are nx wide, the gains n wide) from finite-difference noise"""
    capi = gpu
    from test_dynamics_parity import DERIV_SEQS, _upload_traj
    model, spec, o = make(name, T, fd_mode=0)
    x0, us, xs = initial_trajectory(o, model, seed=31, u_sigma=0.05 if name.startswith("chain6") else 0.3)
    rng = np.random.default_rng(32)
    mults = o.alloc_affine(o.Etot)
    mults["origin"][:] = xs[:T * o.nx]
    mults["jac"][:o.Etot * o.n] = 0.01 * rng.normal(size=o.Etot * o.n)
    mu = 100.0
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        _upload_traj(ctx, xs, us)
        for k, s in (("origin", "MULT_ORIGIN"), ("val", "MULT_VAL"), ("jac", "MULT_JAC")):
            if ctx.seq_size(s):
                ctx.upload(s, mults[k][:ctx.seq_size(s)], 0, 1)
        ctx.linearize()
        d = o.alloc_derivs()
        for key, seq in DERIV_SEQS.items():
            sz = ctx.seq_size(seq)
            if sz:
                d[key][:sz] = ctx.download(seq, 0, 1)[0]
        rc, reg, mu_out, _ = ctx.backward(0.0, mu)
        ref = o.backward(d, xs, mults, reg=0.0, mu=mu)
        assert reg[0] == ref["reg"] and mu_out[0] == ref["mu"]
        assert rel_err(ctx.download("FB_VAL", 0, 1)[0], ref["fb"]["val"][:T * o.m]) < 1e-10
        assert rel_err(ctx.download("FB_JAC", 0, 1)[0], ref["fb"]["jac"][:T * o.m * o.n]) < 1e-10
        assert np.array_equal(ctx.download("FB_ORIGIN", 0, 1)[0], xs[:T * o.nx])


@pytest.mark.gpu
@pytest.mark.parametrize("name,fd_mode,tol", [("chain6ff", 0, 1e-6), ("chain6ff", 2, 1e-3), ("tree38ff", 0, 1e-6),
                                              ("chain6ff_frame", 0, 1e-6), ("tree38ff_frame", 0, 1e-5)])
def test_free_flyer_whole_solve(gpu, name, fd_mode, tol):
    """solve<M> on a free-floating mechanism, device against oracle: same iteration count, same mu / reg history, same
    trajectory to the finite-difference noise of the jacobians.  Unconstrained, the cost c/2 |u|^2 is minimised by u = 0 in one
    iteration (plumbing: every sequence operation runs with nx = n + 1).  With the frame constraint the reference's constraint
    jacobian (the top rows of the WORLD-frame spatial jacobian, DESIGN.md "reference quirks") is not dp/dq, the direction is not a
    descent direction and the line search runs to its floor on both sides: the logs and the (unchanged) iterates must still agree"""
    from ddp_pinocchio_amd import solver
    capi = gpu
    T, B, iters, thr, mu, w, n = 8, 2, 6, 1e-9, 1e2, 1e-1, 10.0
    model, spec, o = make(name, T, batch=B, fd_mode=fd_mode)
    inits, refs = [], []
    for b in range(B):
        us0 = (0.02 if name.startswith("chain6") else 0.2) * np.random.default_rng(50 + b).normal(size=T * model.nv)
        xs0 = o.rollout(neutral_state(model), us0)
        inits.append((xs0, us0))
        refs.append(o.solve(xs0, us0, np.zeros(max(1, o.Etot * o.n)), max_iterations=iters, threshold=thr, mu=mu, reg=0.0, w=w, n=n))
    with capi.Context(spec, flags=0 if fd_mode else capi.FLAG_NO_TENSORS) as ctx:
        for b, (xs0, us0) in enumerate(inits):
            ctx.upload("X", xs0, b, 1); ctx.upload("U", us0, b, 1); ctx.upload("X_NEW", xs0, b, 1); ctx.upload("U_NEW", us0, b, 1)
            if o.Etot:
                ctx.upload("MULT_ORIGIN", xs0[:T * o.nx], b, 1)
                ctx.upload("MULT_VAL", np.zeros(o.Etot), b, 1)
                ctx.upload("MULT_JAC", np.zeros(o.Etot * o.n), b, 1)
        log = solver.solve(ctx, iters, thr, mu, 0.0, w, n)
        xs, us = ctx.download("X"), ctx.download("U")
    for b in range(B):
        xs_ref, us_ref, fb_ref, log_ref = refs[b]
        assert int(log["iterations"][b]) == log_ref["iterations"]
        assert log["mu"][b] == log_ref["mu"] and log["reg"][b] == log_ref["reg"]
        assert float(np.max(np.abs(xs[b] - xs_ref))) < tol, (b, float(np.max(np.abs(xs[b] - xs_ref))))
        assert float(np.max(np.abs(us[b] - us_ref))) < 10 * tol * max(1.0, float(np.max(np.abs(us_ref)))), b
        assert float(log["last_step"][b]) == log_ref["last_step"]
        assert abs(log["opt_constr"][b] - log_ref["opt_constr"]) <= 100 * tol * max(1.0, log_ref["opt_constr"])
