"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the numpy restatement):
the C oracle on CPU, and the HIP path on the GPU, must both reproduce them."""
import glob
import os

import numpy as np
import pytest

from synth import rel_err, synth_sweep_inputs, upload_sweep_inputs

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _inputs(g):
    nv, T, ne = int(g["nv"]), int(g["T"]), [int(v) for v in g["ne"]]
    indef = int(g["indefinite_at"])
    d, xs, us, mults = synth_sweep_inputs(T, nv, ne, seed=int(g["seed"]), indefinite_at=None if indef < 0 else indef)
    if "xs" in g:   # the small cases carry their inputs verbatim: the regenerated ones must be the same bits
        assert np.array_equal(xs, g["xs"]) and np.array_equal(d["fx"], g["d_fx"]) and np.array_equal(mults["jac"], g["mult_jac"])
    return nv, T, ne, d, xs, us, mults


def _check(g, T, n, m, k, K, Vx, Vxx, tol):
    assert rel_err(k.reshape(T, m), g["k"]) < tol and rel_err(Vx.reshape(T, n), g["Vx"]) < tol
    Kf = np.array([np.linalg.norm(K[t * m * n:(t + 1) * m * n]) for t in range(T)])
    Vf = np.array([np.linalg.norm(Vxx[t * n * n:(t + 1) * n * n]) for t in range(T)])
    assert rel_err(Kf, g["K_fro"]) < tol and rel_err(Vf, g["Vxx_fro"]) < tol
    if "K" in g:
        Kg = np.concatenate([g["K"][t].ravel(order="F") for t in range(T)])
        assert rel_err(K[:T * m * n], Kg) < tol


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_c_oracle_reproduces_golden(path):
    from ddp_pinocchio_amd import capi
    from oracle.binding import Oracle
    g = np.load(path)
    nv, T, ne, d, xs, us, mults = _inputs(g)
    model = capi.BuiltinModel(capi.BUILTIN_PENDULUM)
    model.nv = nv
    r = Oracle(model, T, ne=ne).backward(d, xs, mults, float(g["reg_in"]), float(g["mu_in"]))
    assert r["restarts"] == int(g["restarts"]) and r["reg"] == float(g["reg_out"]) and r["mu"] == float(g["mu_out"])
    _check(g, T, 2 * nv, nv, r["fb"]["val"][:T * nv], r["fb"]["jac"], r["Vx"], r["Vxx"], 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("path", [p for p in GOLD if "nv3" not in p], ids=lambda p: os.path.basename(p)[:-4])
def test_hip_reproduces_golden(gpu, path):
    capi = gpu
    g = np.load(path)
    nv, T, ne, d, xs, us, mults = _inputs(g)
    which = {1: capi.BUILTIN_PENDULUM, 6: capi.BUILTIN_CHAIN6, 38: capi.BUILTIN_TREE38}[nv]
    ne_a = np.asarray(ne, dtype=np.int64)
    spec = capi.ProblemSpec(capi.BuiltinModel(which, 1), T, eq_kind=capi.EQ_CONFIG if ne_a.sum() else capi.EQ_NONE,
                            ne=ne_a, eq_target=np.zeros(int(ne_a.sum())))
    with capi.Context(spec, flags=capi.FLAG_TRACE) as ctx:
        upload_sweep_inputs(ctx, d, xs, us, mults, 0)
        rc, reg, mu, restarts = ctx.backward(float(g["reg_in"]), float(g["mu_in"]))
        assert restarts[0] == int(g["restarts"]) and reg[0] == float(g["reg_out"]) and mu[0] == float(g["mu_out"])
        _check(g, T, 2 * nv, nv, ctx.download("FB_VAL")[0], ctx.download("FB_JAC")[0], ctx.download("VX_TRACE")[0],
               ctx.download("VXX_TRACE")[0], 1e-10)
