"""Constrained problems at the Talos size in full-DDP mode (second-order tensors of f AND of the constraint chain in the
sweep): the problem class the benchmark's own data never reach (its V is identically zero).  The device
linearises, and the sweep is checked on the device's own derivative inputs against the oracle."""
import numpy as np
import pytest

from problems import make
from synth import rel_err
from test_dynamics_parity import DERIV_SEQS, TENSOR_SEQS


def _device_derivs(ctx, o, b=0):
    d = o.alloc_derivs()
    for k, sname in {**DERIV_SEQS, **TENSOR_SEQS}.items():
        sz = ctx.seq_size(sname)
        if sz:
            d[k][:sz] = ctx.download(sname, b, 1)[0]
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,jac_sigma,mu", [("tree38_config", 30, 0.01, 1e3), ("tree38_frame", 12, 0.01, 1e3)])
def test_constrained_full_ddp_sweep_talos(gpu, name, T, jac_sigma, mu):
    capi = gpu
    model, spec, o = make(name, T, fd_mode=2)
    m, nx = o.m, o.nx
    us = 0.1 * np.random.default_rng(5000).normal(size=T * m)
    xs = o.rollout(np.zeros(nx), us)
    mults = o.alloc_affine(o.Etot)
    mults["origin"][:] = xs[:T * nx]
    mults["jac"][:o.Etot * o.n] = jac_sigma * np.random.default_rng(6000).normal(size=o.Etot * o.n)
    with capi.Context(spec, flags=capi.FLAG_TRACE) as ctx:
        ctx.upload("X", xs); ctx.upload("U", us); ctx.upload("X_NEW", xs); ctx.upload("U_NEW", us)
        for k, sname in (("origin", "MULT_ORIGIN"), ("val", "MULT_VAL"), ("jac", "MULT_JAC")):
            ctx.upload(sname, mults[k][:ctx.seq_size(sname)])
        ctx.linearize()
        d = _device_derivs(ctx, o)
        for k in d:
            assert np.all(np.isfinite(d[k])), k
        ref = o.backward(d, xs, mults, 0.0, mu, max_restarts=40)
        if ref["restarts"] < 0:
            # Q_uu never turns positive definite (frame constraint in tensor mode: the reference's WORLD-frame jacobian rows,
            # DESIGN.md 4d): the reference's while(!success) would spin for ever; both sides must give up the same way
            with pytest.raises(capi.DdpHipError) as exc:
                ctx.backward(0.0, mu, 40)
            assert exc.value.code == capi.E_MAX_RESTARTS
            return
        rc, reg, mu_out, restarts = ctx.backward(0.0, mu, 40)
        assert int(restarts[0]) == ref["restarts"] and mu_out[0] == ref["mu"] and reg[0] == ref["reg"]
        K = ctx.download("FB_JAC")[0]
        k_ = ctx.download("FB_VAL")[0]
        Vx = ctx.download("VX_TRACE")[0]
        assert rel_err(K, ref["fb"]["jac"][:K.size]) < 1e-9 and rel_err(k_, ref["fb"]["val"][:k_.size]) < 1e-9
        assert rel_err(Vx, ref["Vx"]) < 1e-9
        assert float(np.max(np.abs(K))) > 1e-3          # a non-trivial feedback, unlike the unconstrained benchmark data
