"""Round-3 additions at the boundary, on the device against the oracle:
  * ddp_hip_forward with n_alpha = 0: forward_pass(do_linesearch = false), ddp_fwd.ipp:61-63;
  * ddp_hip_shard_pick / ddp_hip_shard_broadcast: the best-cost pick as one collective on resident data and the winner's
    broadcast (kernel map C1 / C2) over a one-rank RCCL communicator and with no communicator (the multi-rank
    ownership / root logic is covered over gloo in tests/test_shard_gloo.py)."""
import numpy as np
import pytest

from problems import initial_trajectory, make
from synth import rel_err


def _setup(capi, name, T, batch, seed, u_sigma, jac_sigma=0.0):
    model, spec, o = make(name, T, batch=batch, fd_mode=0)
    trajs = [initial_trajectory(o, model, seed=seed + b, u_sigma=u_sigma) for b in range(batch)]
    ctx = capi.Context(spec, flags=capi.FLAG_NO_TENSORS)
    rng = np.random.default_rng(seed)
    mults = []
    for b, (x0, us, xs) in enumerate(trajs):
        ctx.upload("X", xs, b, 1); ctx.upload("U", us, b, 1); ctx.upload("X_NEW", xs, b, 1); ctx.upload("U_NEW", us, b, 1)
        m = o.alloc_affine(o.Etot)
        m["origin"][:] = xs[:T * o.nx]
        if o.Etot:
            m["jac"][:o.Etot * o.n] = jac_sigma * rng.normal(size=o.Etot * o.n)
            for k, sname in (("origin", "MULT_ORIGIN"), ("val", "MULT_VAL"), ("jac", "MULT_JAC")):
                ctx.upload(sname, m[k][:ctx.seq_size(sname)], b, 1)
        mults.append(m)
    return model, o, ctx, trajs, mults


@pytest.mark.gpu
def test_forward_without_linesearch_takes_the_full_step(gpu):
    """do_linesearch == false (ddp_fwd.ipp:61-63): the rollout at step 1 is returned whatever the cost does.  The gains are
    scaled up so that the full step INCREASES the cost (a line search would halve): X_NEW / U_NEW must still be the step-1
    rollout of the oracle, and step_out 1."""
    capi = gpu
    T = 12
    model, o, ctx, trajs, mults = _setup(capi, "chain6", T, 1, seed=5, u_sigma=0.05, jac_sigma=0.01)
    with ctx:
        x0, us, xs = trajs[0]
        ctx.linearize()
        rc, reg, mu, _ = ctx.backward(0.0, 100.0)
        # blow the feed-forward term up: the full step overshoots
        k = ctx.download("FB_VAL")[0]
        ctx.upload("FB_VAL", 40.0 * k)
        fb = {"origin": ctx.download("FB_ORIGIN")[0], "val": 40.0 * k, "jac": ctx.download("FB_JAC")[0]}
        dc1, xs1, us1 = o.forward_alpha(1.0, xs, us, mults[0], fb, float(mu[0]))
        assert dc1 > 0, "the test needs a full step that increases the cost"
        rc, step, dcost = ctx.forward(mu, n_alpha=0)
        assert step[0] == 1.0
        assert rel_err(ctx.download("X_NEW")[0], xs1) < 1e-9 and rel_err(ctx.download("U_NEW")[0], us1) < 1e-9
        assert abs(dcost[0] - dc1) <= 1e-9 * max(1.0, abs(dc1))
        # with the line search the same inputs are halved
        rc, step_ls, _ = ctx.forward(mu, n_alpha=8)
        assert step_ls[0] < 1.0


def _total_costs(ctx, dcost):
    return ctx.download("COSTS_OLD").sum(axis=1) + dcost


@pytest.mark.gpu
@pytest.mark.parametrize("with_comm", [False, True])
def test_shard_pick_and_broadcast_single_rank(gpu, with_comm):
    """the device-side pick (pick.hip) + the one-rank RCCL all-gather / grouped broadcast of csrc/comm.cpp"""
    capi = gpu
    T, B = 8, 5
    # a constrained problem: V != 0, so the instances' costs differ (the unconstrained benchmark data give cost 0 for all)
    model, o, ctx, trajs, mults = _setup(capi, "chain6", T, B, seed=31, u_sigma=0.05, jac_sigma=0.01)
    comm = capi.Comm(capi.Comm.unique_id(), 0, 1, 0) if with_comm else None
    try:
        with ctx:
            ctx.linearize()
            rc, reg, mu, _ = ctx.backward(0.0, 100.0)
            rc, step, dcost = ctx.forward(mu, n_alpha=8)
            tot = _total_costs(ctx, dcost)
            cost, idx = ctx.shard_pick(comm)
            j = int(np.argmin(tot))
            assert np.min(np.diff(np.sort(tot))) > 1e-9 * np.max(np.abs(tot)), "the test needs distinct costs"
            assert idx == j and abs(cost - tot[j]) <= 1e-12 * max(1.0, abs(tot[j]))
            # the winner's trajectory and gains into local instance 0 (dst != src unless the winner is 0)
            ctx.swap_traj()
            X, U, K = ctx.download("X"), ctx.download("U"), ctx.download("FB_JAC")
            dst = (j + 1) % B
            ctx.shard_broadcast(idx, dst_local=dst, comm=comm)
            X2, U2, K2 = ctx.download("X"), ctx.download("U"), ctx.download("FB_JAC")
            assert np.array_equal(X2[dst], X[j]) and np.array_equal(U2[dst], U[j]) and np.array_equal(K2[dst], K[j])
            for b in range(B):
                if b != dst:
                    assert np.array_equal(X2[b], X[b]) and np.array_equal(K2[b], K[b])
    finally:
        if comm is not None:
            comm.close()


@pytest.mark.gpu
def test_backward_reads_one_of_each_mirrored_half_slab_bit_for_bit(gpu, monkeypatch):
    """Mode-2 tensors are symmetric in their two input indices bit for bit (the stencil writes one value to both entries,
    problem.hpp:283-292).  K3 then skips the half-slab f_xx(:, 0:m, c) of the columns c >= m and takes its contraction from
    the mirror image (bwd_split.h, job kind 2): the sweep must give the very same gains and value function, bit for bit, as
    with every half-slab read (DDP_HIP_K3_NO_SYM=1) -- and uploaded tensors (no symmetry known) take the full path."""
    capi = gpu
    T = 5
    model, spec, o = make("tree38", T, batch=2, fd_mode=2)
    with capi.Context(spec, flags=capi.FLAG_TRACE) as ctx:
        for b in range(2):
            x0, us, xs = initial_trajectory(o, model, seed=60 + b, u_sigma=0.4)
            ctx.upload("X", xs, b, 1); ctx.upload("U", us, b, 1)
        ctx.linearize()
        # a V_x that is not zero: terminal cost gradient
        ctx.upload("LFX", np.random.default_rng(1).normal(size=(2, o.n)))
        ctx.upload("LFXX", np.tile(np.eye(o.n).reshape(-1), (2, 1)))
        fxx = ctx.download("FXX")[0].reshape(T, o.n, o.n, o.n)                   # [t][k][j][i]
        assert np.array_equal(fxx, fxx.transpose(0, 2, 1, 3)), "mode-2 f_xx is symmetric bit for bit"

        def sweep():
            rc, reg, mu, rs = ctx.backward(0.0, 1.0)
            return ctx.download("FB_JAC"), ctx.download("FB_VAL"), ctx.download("VX_TRACE"), ctx.download("VXX_TRACE"), rs
        sym = sweep()
        monkeypatch.setenv("DDP_HIP_K3_NO_SYM", "1")
        full = sweep()
        monkeypatch.delenv("DDP_HIP_K3_NO_SYM")
        for a_, b_ in zip(sym, full):
            assert np.array_equal(a_, b_)
        assert float(np.max(np.abs(sym[2]))) > 0
        # tensors from outside: the flag drops (the same values here, so the answer is still the same)
        ctx.upload("FXX", ctx.download("FXX"))
        again = sweep()
        for a_, b_ in zip(sym, again):
            assert np.array_equal(a_, b_)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tree38_config", "tree38_frame", "tree38ff", "tree38ff_frame"])
def test_constrained_and_free_flyer_problems_take_the_latency_forward_kernel(gpu, name):
    """forward_kernel_lat2 was `<38>`-only and unconstrained; constrained problems and free-flyer models fell
    to the one-lane-per-rollout kernel (365 / 300 ms per pass at 64 seeds against 10).  Now the rollout of every Talos-size
    tree runs on the latency kernel (free-flyer root: rbd::aba_tree_coop2w<FF>, SE(3) difference / integrate in the kernel;
    constraints: the candidates' cost terms on cand_cost_kernel, parallel over t) -- and lands on the oracle's step."""
    capi = gpu
    T = 8
    model, spec, o = make(name, T, batch=1, fd_mode=0)
    x0, us, xs = initial_trajectory(o, model, seed=17, u_sigma=0.3)
    mults = o.alloc_affine(o.Etot)
    mults["origin"][:] = xs[:T * o.nx]
    if o.Etot:
        mults["jac"][:o.Etot * o.n] = 0.01 * np.random.default_rng(2).normal(size=o.Etot * o.n)
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        assert ctx.info()["fwd_path"] == 1
        ctx.upload("X", xs); ctx.upload("U", us); ctx.upload("X_NEW", xs); ctx.upload("U_NEW", us)
        for k, sname in (("origin", "MULT_ORIGIN"), ("val", "MULT_VAL"), ("jac", "MULT_JAC")):
            if ctx.seq_size(sname):
                ctx.upload(sname, mults[k][:ctx.seq_size(sname)])
        ctx.linearize()
        rc, reg, mu, _ = ctx.backward(0.0, 100.0)
        fb = {"origin": ctx.download("FB_ORIGIN")[0], "val": ctx.download("FB_VAL")[0], "jac": ctx.download("FB_JAC")[0]}
        step_ref, xs_ref, us_ref, _ = o.forward(xs, us, mults, fb, float(mu[0]))
        rc, step, dcost = ctx.forward(mu, n_alpha=8)
        assert step[0] == step_ref
        assert rel_err(ctx.download("X_NEW")[0], xs_ref) < 1e-8 and rel_err(ctx.download("U_NEW")[0], us_ref) < 1e-8


@pytest.mark.gpu
def test_k3h_bit_for_bit_at_size(gpu, monkeypatch):
    """the same at BASELINE's size: T = 200, the benchmark's inputs (x0 neutral, u ~ N(0, 0.1^2)) plus a terminal gradient so
    that V_x != 0 along the whole horizon; K3h against the dense kernel at every step of four instances"""
    capi = gpu
    T, B = 200, 4
    model, spec, o = make("tree38", T, batch=B, fd_mode=2)
    with capi.Context(spec, flags=capi.FLAG_TRACE) as ctx:
        us = np.stack([0.1 * np.random.default_rng(0xDD9000 + 3000 + g).normal(size=T * o.m) for g in range(B)])
        ctx.upload("X", np.zeros((B, (T + 1) * o.nx))); ctx.upload("U", us)
        ctx.rollout()
        ctx.linearize()
        ctx.upload("LFX", 0.1 * np.random.default_rng(7).normal(size=(B, o.n)))
        ctx.upload("LFXX", np.tile(np.eye(o.n).reshape(-1), (B, 1)))
        assert ctx.bwd_stream_bytes() < 0.4 * 8 * (o.n ** 3 + o.n * o.n * o.m + o.n * o.m * o.m)     # K3h is what runs

        def sweep():
            rc, reg, mu, rs = ctx.backward(0.0, 1.0, 8)
            return ctx.download("FB_JAC"), ctx.download("FB_VAL"), ctx.download("VX_TRACE"), rs, reg, mu
        half = sweep()
        monkeypatch.setenv("DDP_HIP_K3_NO_SYM", "1")
        dense = sweep()
        monkeypatch.delenv("DDP_HIP_K3_NO_SYM")
        for a_, b_ in zip(half, dense):
            assert np.array_equal(a_, b_)
        assert np.all(np.isfinite(half[0])) and float(np.max(np.abs(half[2]))) > 0


@pytest.mark.gpu
def test_k3h_on_analytic_mode1_tensors_bit_for_bit(gpu, monkeypatch):
    """Analytic mode 1 (lin_analytic.hip) leaves tensors with a structure of its own: the configuration rows of every f_xx / f_ux
    column are zeros it wrote itself (q+ = q + dt v has constant jacobian rows) and f_uu is zero (M^-1 does not depend on u), but
    f_xx is NOT symmetric (a forward difference of jacobians).  K3h then reads the lower half of every column and nothing of
    f_uu (half_mode 2): the sweep must match the dense kernel (DDP_HIP_K3_NO_HALF=1) bit for bit, and uploaded tensors must
    drop the assumption."""
    capi = gpu
    T, B = 12, 3
    model, spec, o = make("tree38", T, batch=B, fd_mode=1, first_order_fd=0)
    full_bytes = 8 * (o.n ** 3 + o.n * o.n * o.m + o.n * o.m * o.m)
    with capi.Context(spec, flags=capi.FLAG_TRACE) as ctx:
        us = np.stack([0.1 * np.random.default_rng(500 + g).normal(size=T * o.m) for g in range(B)])
        ctx.upload("X", np.zeros((B, (T + 1) * o.nx))); ctx.upload("U", us)
        ctx.rollout()
        assert ctx.bwd_stream_bytes() == full_bytes                               # nothing known about the tensors yet
        ctx.linearize()
        assert ctx.bwd_stream_bytes() == 8 * (o.n * o.n + o.n * o.m) * (o.n - o.m)
        fxx = ctx.download("FXX")[0].reshape(T, o.n, o.n, o.n)                    # [t][i][j][k]
        fux = ctx.download("FUX")[0].reshape(T, o.n, o.m, o.n)
        assert not np.any(fxx[..., :o.m]) and not np.any(fux[..., :o.m]) and not np.any(ctx.download("FUU"))
        assert not np.array_equal(fxx, fxx.transpose(0, 2, 1, 3))
        ctx.upload("LFX", 0.1 * np.random.default_rng(7).normal(size=(B, o.n)))
        ctx.upload("LFXX", np.tile(np.eye(o.n).reshape(-1), (B, 1)))

        def sweep():
            rc, reg, mu, rs = ctx.backward(0.0, 1.0, 8)
            return ctx.download("FB_JAC"), ctx.download("FB_VAL"), ctx.download("VX_TRACE"), rs, reg, mu
        half = sweep()
        monkeypatch.setenv("DDP_HIP_K3_NO_HALF", "1")
        dense = sweep()
        monkeypatch.delenv("DDP_HIP_K3_NO_HALF")
        for a_, b_ in zip(half, dense):
            assert np.array_equal(a_, b_)
        assert np.all(np.isfinite(half[0])) and float(np.max(np.abs(half[2]))) > 0
        ctx.upload("FUX", ctx.download("FUX"))                                    # tensors from outside: the full read again
        assert ctx.bwd_stream_bytes() == full_bytes
        again = sweep()
        for a_, b_ in zip(half, again):
            assert np.array_equal(a_, b_)
