"""Backward-sweep parity: HIP path (through the C-ABI) vs the CPU oracle on identical derivative inputs.

Bar: bit-exact flat indexing; <= 1e-10 relative on V_x, V_xx, k, K at every step;
identical LLT-failure / restart decisions (reg, mu bit-exact).
"""
import numpy as np
import pytest

from synth import rel_err, synth_sweep_inputs, upload_sweep_inputs

TOL = 1e-10   # north_star: within 1e-10 rel on V_x / K (double)


def _oracle(nv, T, ne):
    from ddp_pinocchio_amd import capi
    from oracle.binding import Oracle
    model = capi.BuiltinModel(capi.BUILTIN_PENDULUM)
    model.nv = nv
    return Oracle(model, T, ne=ne)


def _spec(capi, nv, T, ne, batch):
    if nv == 1:
        model = capi.BuiltinModel(capi.BUILTIN_PENDULUM)
    elif nv == 6:
        model = capi.BuiltinModel(capi.BUILTIN_CHAIN6)
    elif nv == 38:
        model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
    else:
        raise ValueError(nv)
    ne = np.asarray(ne, dtype=np.int64)
    kind = capi.EQ_CONFIG if ne.sum() else capi.EQ_NONE
    return capi.ProblemSpec(model, T, batch=batch, eq_kind=kind, ne=ne, eq_target=np.zeros(int(ne.sum())))


def _compare_instance(ctx, b, o, ref, T):
    n, m, nx = o.n, o.m, o.nx
    k = ctx.download("FB_VAL", b, 1)[0]
    K = ctx.download("FB_JAC", b, 1)[0]
    org = ctx.download("FB_ORIGIN", b, 1)[0]
    Vx = ctx.download("VX_TRACE", b, 1)[0]
    Vxx = ctx.download("VXX_TRACE", b, 1)[0]
    worst = 0.0
    for t in range(T):
        for got, exp, w in ((k, ref["fb"]["val"], m), (K, ref["fb"]["jac"], m * n), (Vx, ref["Vx"], n), (Vxx, ref["Vxx"], n * n)):
            worst = max(worst, rel_err(got[t * w:(t + 1) * w], exp[t * w:(t + 1) * w]))
    assert np.array_equal(org, ref["fb"]["origin"][:T * nx])   # plain copies: bit exact
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("nv,T,ne_kind,tensors", [
    (1, 50, "last", True),      # pendulum shape (config P): e = 1 at t = T-2
    (1, 7, "none", True),
    (6, 10, "all", True),       # UR5-like shape (config U): e = 6 every step, reference horizon 10
    (6, 100, "all", True),      # ... BASELINE horizon 100
    (6, 16, "none", False),     # tensor-free (Gauss-Newton) variant
    (38, 8, "none", True),      # Talos-like shape at the real n, m with tensors, short horizon
    (38, 6, "last", True),
])
def test_backward_parity(gpu, nv, T, ne_kind, tensors):
    capi = gpu
    e = {"last": [0] * (T - 2) + [nv, 0], "all": [nv] * T, "none": [0] * T}[ne_kind]
    o = _oracle(nv, T, e)
    batch = 3
    flags = capi.FLAG_TRACE | (0 if tensors else capi.FLAG_NO_TENSORS)
    with capi.Context(_spec(capi, nv, T, e, batch), flags=flags) as ctx:
        refs = []
        for b in range(batch):
            d, xs, us, mults = synth_sweep_inputs(T, nv, e, seed=1000 + 17 * b + nv, tensors=tensors)
            upload_sweep_inputs(ctx, d, xs, us, mults, b, tensors=tensors)
            refs.append(o.backward(d, xs, mults, reg=0.0, mu=10.0))
        rc, reg, mu, restarts = ctx.backward(reg=0.0, mu=10.0)
        assert rc == 0 and not restarts.any()
        for b in range(batch):
            assert refs[b]["restarts"] == 0
            assert reg[b] == refs[b]["reg"] and mu[b] == refs[b]["mu"]
            worst = _compare_instance(ctx, b, o, refs[b], T)
            assert worst < TOL, (b, worst)


@pytest.mark.gpu
def test_backward_restart_decisions(gpu):
    # instance 1 has an indefinite Q_uu at t = 3 -> restarts with the reg/mu rule of ddp_bwd.ipp:105-110;
    # instances 0 and 2 must be untouched by its restart
    capi = gpu
    nv, T = 6, 8
    e = [0] * T
    o = _oracle(nv, T, e)
    with capi.Context(_spec(capi, nv, T, e, 3), flags=capi.FLAG_TRACE) as ctx:
        refs = []
        for b in range(3):
            d, xs, us, mults = synth_sweep_inputs(T, nv, e, seed=50 + b, indefinite_at=3 if b == 1 else None)
            upload_sweep_inputs(ctx, d, xs, us, mults, b)
            refs.append(o.backward(d, xs, mults, reg=0.0, mu=0.25))
        rc, reg, mu, restarts = ctx.backward(reg=0.0, mu=0.25)
        assert rc == capi.EV_LLT_RESTART
        assert refs[1]["restarts"] >= 1
        for b in range(3):
            assert restarts[b] == refs[b]["restarts"]
            assert reg[b] == refs[b]["reg"] and mu[b] == refs[b]["mu"]     # bit exact decisions
            assert _compare_instance(ctx, b, o, refs[b], T) < TOL


@pytest.mark.gpu
def test_backward_max_restarts(gpu):
    capi = gpu
    nv, T = 1, 5
    e = [0] * T
    with capi.Context(_spec(capi, nv, T, e, 1), flags=capi.FLAG_TRACE) as ctx:
        d, xs, us, mults = synth_sweep_inputs(T, nv, e, seed=3)
        d["luu"][:] = -1e30   # never positive definite within the bound
        upload_sweep_inputs(ctx, d, xs, us, mults, 0)
        with pytest.raises(capi.DdpHipError) as ei:
            ctx.backward(reg=0.0, mu=1.0, max_restarts=3)
        assert ei.value.code == capi.E_MAX_RESTARTS


@pytest.mark.gpu
def test_split_and_generic_kernels_agree(gpu, monkeypatch):
    """The Talos-like shape normally runs the split K3/K4 kernels (compile-time shape, MFMA dense product); forcing
    the run-time-shaped pair on the same inputs must give the same sweep (two implementations, one oracle)."""
    capi = gpu
    nv, T = 38, 5
    e = [0] * T
    o = _oracle(nv, T, e)
    d, xs, us, mults = synth_sweep_inputs(T, nv, e, seed=77)
    ref = o.backward(d, xs, mults, reg=0.0, mu=10.0)
    worst = {}
    for mode in ("split", "generic"):
        if mode == "generic":
            monkeypatch.setenv("DDP_HIP_GENERIC_BWD", "1")
        else:
            monkeypatch.delenv("DDP_HIP_GENERIC_BWD", raising=False)
        with capi.Context(_spec(capi, nv, T, e, 1), flags=capi.FLAG_TRACE) as ctx:
            upload_sweep_inputs(ctx, d, xs, us, mults, 0)
            rc, reg, mu, restarts = ctx.backward(reg=0.0, mu=10.0)
            assert rc == 0
            worst[mode] = _compare_instance(ctx, 0, o, ref, T)
    assert worst["split"] < TOL and worst["generic"] < TOL, worst


@pytest.mark.gpu
def test_backward_parity_full_size(gpu):
    """BASELINE.json's full size: Talos-like n = 76, m = 38, horizon T = 200, with all three second-order tensors
    (1.23 GB of derivative inputs for the one instance), directly against the C oracle, plus two size-independent
    properties of the sweep: (i) with the tensors zeroed it equals the tensor-free (Gauss-Newton) context bit for bit,
    (ii) every instance of a batch is independent of its neighbours (same inputs -> same bits in every slot)."""
    capi = gpu
    nv, T = 38, 200
    e = [0] * T
    o = _oracle(nv, T, e)
    d, xs, us, mults = synth_sweep_inputs(T, nv, e, seed=2024)
    ref = o.backward(d, xs, mults, reg=0.0, mu=10.0)
    assert ref["restarts"] == 0
    with capi.Context(_spec(capi, nv, T, e, 2), flags=capi.FLAG_TRACE) as ctx:
        for b in range(2):
            upload_sweep_inputs(ctx, d, xs, us, mults, b)
        rc, reg, mu, restarts = ctx.backward(reg=0.0, mu=10.0)
        assert rc == 0 and not restarts.any()
        assert _compare_instance(ctx, 0, o, ref, T) < TOL
        K = ctx.download("FB_JAC")
        Vxx = ctx.download("VXX_TRACE")
        assert np.array_equal(K[0], K[1]) and np.array_equal(Vxx[0], Vxx[1])          # (ii)
        # (i) zero tensors == tensor-free context
        for s in ("FXX", "FUX", "FUU"):
            ctx.fill(s, 0.0)
        ctx.backward(reg=0.0, mu=10.0)
        K0 = ctx.download("FB_JAC", 0, 1)[0]
    with capi.Context(_spec(capi, nv, T, e, 1), flags=capi.FLAG_TRACE | capi.FLAG_NO_TENSORS) as ctx:
        upload_sweep_inputs(ctx, d, xs, us, mults, 0, tensors=False)
        ctx.backward(reg=0.0, mu=10.0)
        Kgn = ctx.download("FB_JAC", 0, 1)[0]
    for k in ("fxx", "fux", "fuu"):
        d[k][:] = 0.0
    ref0 = o.backward(d, xs, mults, reg=0.0, mu=10.0)
    assert np.array_equal(K0, Kgn)
    assert rel_err(Kgn, ref0["fb"]["jac"][:Kgn.size]) < TOL


@pytest.mark.gpu
def test_shard_best_single_rank_rccl(gpu):
    """ddp_hip_shard_best over a one-rank RCCL communicator (the multi-rank reduction logic is covered on gloo)"""
    import ctypes as C
    capi = gpu
    L = capi.lib()
    uid = (C.c_ubyte * 128)()
    assert L.ddp_hip_comm_unique_id(uid) == 0
    comm = C.c_void_p()
    assert L.ddp_hip_comm_init(uid, 0, 1, 0, C.byref(comm)) == 0
    cost, idx = C.c_double(), C.c_int64()
    assert L.ddp_hip_shard_best(comm, -2.5, 17, C.byref(cost), C.byref(idx)) == 0
    assert cost.value == -2.5 and idx.value == 17
    assert L.ddp_hip_comm_destroy(comm) == 0


@pytest.mark.gpu
def test_two_contexts_from_two_host_threads(gpu):
    """include/ddp_hip/ddp_hip.h: contexts are independent and may be driven from different host threads (one context =
    one HIP stream).  Two contexts running a full iteration concurrently give bit for bit what each gives alone."""
    import threading
    from problems import initial_trajectory, make
    capi = gpu
    T, B = 6, 2
    model, spec, o = make("tree38", T, batch=B, fd_mode=2)

    def setup(seed):
        ctx = capi.Context(spec)
        for b in range(B):
            x0, us, xs = initial_trajectory(o, model, seed=seed + b, u_sigma=0.3)
            ctx.upload("X", xs, b, 1); ctx.upload("U", us, b, 1)
            ctx.upload("X_NEW", xs, b, 1); ctx.upload("U_NEW", us, b, 1)
        return ctx

    def iterate(ctx, out, key):
        ctx.linearize()
        rc, reg, mu, restarts = ctx.backward(0.0, 1.0)
        rc2, step, dcost = ctx.forward(mu, n_alpha=8)
        out[key] = (ctx.download("FB_JAC"), ctx.download("X_NEW"), step.copy(), reg.copy())

    alone, together = {}, {}
    for k, seed in (("a", 100), ("b", 200)):
        with setup(seed) as ctx:
            iterate(ctx, alone, k)
    ca, cb = setup(100), setup(200)
    try:
        th = [threading.Thread(target=iterate, args=(ca, together, "a")), threading.Thread(target=iterate, args=(cb, together, "b"))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        ca.close(); cb.close()
    for k in ("a", "b"):
        assert k in together, "a worker thread failed"
        for x, y in zip(alone[k], together[k]):
            assert np.array_equal(x, y)
