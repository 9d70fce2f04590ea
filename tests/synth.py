"""Seeded synthetic inputs for the sweep parity tests (shared by CPU and GPU tests).

The backward sweep is checked *given identical derivative inputs* (SURVEY.md D1): these generators make
derivative sets of the right shapes (reference flat layout) without going through any dynamics.
"""
import numpy as np


def deriv_sizes(T, n, m, nx, Etot):
    return {
        "lfx": n, "lfxx": n * n, "lx": T * n, "lu": T * m, "lxx": T * n * n, "lux": T * m * n, "luu": T * m * m,
        "f_val": T * nx, "fx": T * n * n, "fu": T * n * m, "fxx": T * n ** 3, "fux": T * n * m * n, "fuu": T * n * m * m,
        "eq_val": Etot, "eq_x": Etot * n, "eq_u": Etot * m, "eq_xx": Etot * n * n, "eq_ux": Etot * m * n,
        "eq_uu": Etot * m * m,
    }


def _spd(rng, k, scale):
    a = rng.normal(size=(k, k)) / np.sqrt(k)
    return scale * (a @ a.T)


def synth_sweep_inputs(T, nv, ne, seed, tensors=True, tensor_scale=0.02, luu_diag=1.0, indefinite_at=None):
    """Returns (d, xs, us, mults) as flat float64 arrays in the reference layout."""
    rng = np.random.default_rng(seed)
    n, m, nx = 2 * nv, nv, 2 * nv
    ne = np.asarray(ne, dtype=np.int64)
    Etot = int(ne.sum())
    sz = deriv_sizes(T, n, m, nx, Etot)
    d = {k: np.zeros(max(v, 1)) for k, v in sz.items()}
    d["lfx"][:n] = 0.1 * rng.normal(size=n)
    d["lfxx"][:n * n] = (_spd(rng, n, 0.5) + 0.01 * rng.normal(size=(n, n))).ravel(order="F")
    d["lx"][:T * n] = 0.1 * rng.normal(size=T * n)
    d["lu"][:T * m] = 0.1 * rng.normal(size=T * m)
    for t in range(T):
        d["lxx"][t * n * n:(t + 1) * n * n] = _spd(rng, n, 0.1).ravel(order="F")
        d["lux"][t * m * n:(t + 1) * m * n] = 0.02 * rng.normal(size=m * n)
        diag = luu_diag if (indefinite_at is None or t != indefinite_at) else -0.5 * luu_diag
        d["luu"][t * m * m:(t + 1) * m * m] = (diag * np.eye(m) + _spd(rng, m, 0.05)).ravel(order="F")
        d["fx"][t * n * n:(t + 1) * n * n] = (np.eye(n) + 0.05 * rng.normal(size=(n, n)) / np.sqrt(n)).ravel(order="F")
        d["fu"][t * n * m:(t + 1) * n * m] = (0.3 * rng.normal(size=(n, m)) / np.sqrt(n)).ravel(order="F")
    d["f_val"][:] = rng.normal(size=d["f_val"].size)
    if tensors:
        for k in ("fxx", "fux", "fuu"):
            d[k][:sz[k]] = tensor_scale * rng.normal(size=sz[k]) / n
        for k in ("eq_xx", "eq_ux", "eq_uu"):
            d[k][:sz[k]] = tensor_scale * rng.normal(size=sz[k]) / n
    d["eq_val"][:Etot] = 0.1 * rng.normal(size=Etot)
    d["eq_x"][:Etot * n] = rng.normal(size=Etot * n) / np.sqrt(n)
    d["eq_u"][:Etot * m] = rng.normal(size=Etot * m) / np.sqrt(n)
    xs = rng.normal(size=(T + 1) * nx)
    us = 0.1 * rng.normal(size=T * m)
    mults = {"origin": xs[:T * nx].copy(), "val": np.zeros(max(Etot, 1)), "jac": np.zeros(max(Etot * n, 1))}
    mults["val"][:Etot] = 0.1 * rng.normal(size=Etot)
    mults["jac"][:Etot * n] = 0.1 * rng.normal(size=Etot * n)
    return d, xs, us, mults


def upload_sweep_inputs(ctx, d, xs, us, mults, instance=0, tensors=True):
    names = {"lfx": "LFX", "lfxx": "LFXX", "lx": "LX", "lu": "LU", "lxx": "LXX", "lux": "LUX", "luu": "LUU",
             "f_val": "F_VAL", "fx": "FX", "fu": "FU", "eq_val": "EQ_VAL", "eq_x": "EQ_X", "eq_u": "EQ_U"}
    if tensors:
        names.update({"fxx": "FXX", "fux": "FUX", "fuu": "FUU", "eq_xx": "EQ_XX", "eq_ux": "EQ_UX", "eq_uu": "EQ_UU"})
    for k, s in names.items():
        sz = ctx.seq_size(s)
        if sz:
            ctx.upload(s, d[k][:sz], instance, 1)
    ctx.upload("X", xs, instance, 1)
    ctx.upload("U", us, instance, 1)
    for k, s in (("origin", "MULT_ORIGIN"), ("val", "MULT_VAL"), ("jac", "MULT_JAC")):
        sz = ctx.seq_size(s)
        if sz:
            ctx.upload(s, mults[k][:sz], instance, 1)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    scale = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b)) / scale) if a.size else 0.0


def stepwise_backward_check(oracle_factory, o, d, xs, mults, reg, mu, Vx_tr, Vxx_tr, k_dev, K_dev, ts):
    """Per-step parity of the backward recursion, immune to the conditioning of the whole sweep: for every t in `ts` the
    oracle redoes step t alone (ddp_bwd.ipp:61-146) from the DEVICE's own V_x(t+1), V_xx(t+1) (its trace) and must land on
    the device's k_t, K_t, V_x(t), V_xx(t).  oracle_factory(ne_t, target_t) builds a one-step oracle of the same problem.
    Returns the largest relative error seen over (k, K, V_x, V_xx)."""
    T, n, m, nx = o.T, o.n, o.m, o.nx
    Epre = np.concatenate([[0], np.cumsum(o.ne)]).astype(np.int64)
    worst = 0.0
    for t in ts:
        e, E0 = int(o.ne[t]), int(Epre[t])
        o1 = oracle_factory(t)
        d1 = o1.alloc_derivs()
        if t == T - 1:
            d1["lfx"][:n] = d["lfx"][:n]; d1["lfxx"][:n * n] = d["lfxx"][:n * n]
        else:
            d1["lfx"][:n] = Vx_tr[(t + 1) * n:(t + 2) * n]; d1["lfxx"][:n * n] = Vxx_tr[(t + 1) * n * n:(t + 2) * n * n]
        for key, per in (("lx", n), ("lu", m), ("lxx", n * n), ("lux", m * n), ("luu", m * m), ("f_val", nx), ("fx", n * n), ("fu", n * m),
                         ("fxx", n ** 3), ("fux", n * m * n), ("fuu", n * m * m)):
            if d[key].size >= (t + 1) * per:
                d1[key][:per] = d[key][t * per:(t + 1) * per]
        for key, per in (("eq_val", 1), ("eq_x", n), ("eq_u", m), ("eq_xx", n * n), ("eq_ux", m * n), ("eq_uu", m * m)):
            if e and d[key].size >= (E0 + e) * per:
                d1[key][:e * per] = d[key][E0 * per:(E0 + e) * per]
        m1 = o1.alloc_affine(e)
        m1["origin"][:] = mults["origin"][t * nx:(t + 1) * nx]
        if e:
            m1["val"][:e] = mults["val"][E0:E0 + e]
            m1["jac"][:e * n] = mults["jac"][E0 * n:(E0 + e) * n]
        ref = o1.backward(d1, np.concatenate([xs[t * nx:(t + 1) * nx], xs[(t + 1) * nx:(t + 2) * nx]]), m1, reg, mu)
        assert ref["restarts"] == 0, (t, ref["restarts"])
        worst = max(worst, rel_err(k_dev[t * m:(t + 1) * m], ref["fb"]["val"][:m]), rel_err(K_dev[t * m * n:(t + 1) * m * n], ref["fb"]["jac"][:m * n]),
                    rel_err(Vx_tr[t * n:(t + 1) * n], ref["Vx"][:n]), rel_err(Vxx_tr[t * n * n:(t + 1) * n * n], ref["Vxx"][:n * n]))
    return worst
