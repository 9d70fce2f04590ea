#!/usr/bin/env python3
"""bench.py -- DDP iterations/s on the Talos-like 38-DoF tree, horizon T = 200 (BASELINE.json metric).

One "step" = one full DDP iteration of every resident instance on this GPU:
    linearise (FD f_x, f_u + FD second-order tensors, mode 2)  ->  backward sweep (with tensors)
    ->  forward sweep with 8 batched line-search steps  ->  swap trajectories (+ the reg rule of ddp.hpp:819-824)
with all inputs resident in HBM.  Independent instances (random control seeds) are sharded across
ranks with no data-path collective (weak scaling: `--seeds-per-gpu` instances per GPU); the only exchange
is the best-cost pick: one RCCL all-reduce(min) of 8 bytes + one of the masked index per step.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the library's own stream;
`cpu_baseline` times the CPU oracle (oracle/, a port of the reference algorithm) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seeds-per-gpu", type=int, default=64)
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--mode", choices=["full", "gn"], default="full",
                    help="full = with second-order tensors (the reference's algorithm); gn = tensor-free variant")
    ap.add_argument("--n-alpha", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="development: do not bracket the kernel launches with HIP events (no roofline / per-kernel times)")
    ap.add_argument("--cpu-sample-steps", type=int, default=0, help="horizon of the CPU sample (0 = auto)")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from ddp_pinocchio_amd import capi, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # rehearsal knobs (development only): DDP_BENCH_BACKEND=gloo and DDP_BENCH_SINGLE_DEVICE=1 run the N > 1 code path
    # with every rank on GPU 0 of a one-GPU box; the driver's multi-GPU runs use the defaults (nccl = RCCL, one GPU per rank)
    backend = os.environ.get("DDP_BENCH_BACKEND", "nccl")
    if os.environ.get("DDP_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    T, S = a.horizon, a.seeds_per_gpu
    full = a.mode == "full"
    model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
    spec = capi.ProblemSpec(model, T, dt=0.01, c=1.0, batch=S, fd_mode=2 if full else 0)
    ctx = capi.Context(spec, device=local_rank, flags=0 if full else capi.FLAG_NO_TENSORS)
    nv, nx, m = model.nv, 2 * model.nv, model.nv

    # synthetic inputs: x0 neutral, u_t ~ N(0, 0.1^2), seeded by the GLOBAL instance index
    xs = np.zeros((S, (T + 1) * nx))
    us = np.zeros((S, T * m))
    for s in range(S):
        g = rank * S + s
        us[s] = 0.1 * np.random.default_rng(0xDD9000 + 3000 + g).normal(size=T * m)
    ctx.upload("X", xs)
    ctx.upload("U", us)
    ctx.rollout()
    ctx.upload("X_NEW", ctx.download("X"))
    ctx.upload("U_NEW", us)

    reg = np.zeros(S)
    mu = np.full(S, 1e2)
    phase_ms = {"linearize": 0.0, "backward": 0.0, "forward": 0.0}

    def one_iteration(timed):
        nonlocal reg, mu
        t0 = time.perf_counter()
        ctx.linearize()
        t1 = time.perf_counter()
        rc, reg, mu, restarts = ctx.backward(reg, mu)
        t2 = time.perf_counter()
        rc, step, dcost = ctx.forward(mu, n_alpha=a.n_alpha)
        t3 = time.perf_counter()
        reg = np.where(step >= 0.5, reg / 2, reg)            # ddp.hpp:819-824
        reg = np.where(reg < 1e-5, 0.0, reg)
        ctx.swap_traj()                                      # ddp.hpp:826
        if world > 1:
            # the one exchange step: best-cost pick over all seeds of all ranks (two 8-byte RCCL all-reduces)
            ctx.cost_seq_aug(0, mu)
            costs = ctx.download("COSTS_OLD").sum(axis=1)
            shard.best_of(costs, [rank * S + s for s in range(S)], device=red_dev)
        if timed:
            phase_ms["linearize"] += (t1 - t0) * 1e3
            phase_ms["backward"] += (t2 - t1) * 1e3
            phase_ms["forward"] += (t3 - t2) * 1e3

    for _ in range(a.warmup):
        one_iteration(False)

    # HIP events around every launch of the roofline kernel (K3) and of the few-launch kernels; K4's 200 launches per sweep
    # are left out (an event pair costs stream time): its figure below is the backward phase minus K3
    if a.no_kernel_events:
        ctx.profile_enable(False)
    else:
        ctx.profile_enable(True, kernels=[capi.K_BWD_ASSEMBLE, capi.K_FWD_ROLLOUT, capi.K_LIN_FIRST, capi.K_LIN_SECOND])
    ctx.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_iteration(True)
    ctx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    ctx.profile_enable(False)

    if rank == 0:
        ms_a, n_a = ctx.profile_get(capi.K_BWD_ASSEMBLE)
        ms_g, n_g = ctx.profile_get(capi.K_BWD_GAINS)
        ms_f, n_f = ctx.profile_get(capi.K_FWD_ROLLOUT)
        ms_l1, n_l1 = ctx.profile_get(capi.K_LIN_FIRST)
        ms_l2, n_l2 = ctx.profile_get(capi.K_LIN_SECOND)
        # algorithmic bytes of ONE bwd_contract (K3) launch = one timestep of every resident instance: the three
        # tensors read once (n^3 + n^2 m + n m^2 doubles) + V_x read + the contracted blocks written
        n_, m_ = 2 * nv, nv
        words = (n_ ** 3 + n_ * n_ * m_ + n_ * m_ * m_) + n_ + (n_ * n_ + m_ * n_ + m_ * m_)
        bytes_per_launch = 8.0 * words * S if full else 0.0
        avg_s = (ms_a / max(n_a, 1)) * 1e-3
        achieved = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        out = {
            "metric": "DDP iterations/sec (fwd+bwd sweep), Talos nq=38 T=200",
            "value": world * S * a.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Talos-like 38-DoF tree (nq=nv=38, n=76, m=38), T={T}, {S} seeds/GPU x {a.n_alpha} "
                                   f"line-search alphas, {'full DDP (FD f_x,f_u + FD f_xx,f_ux,f_uu mode 2)' if full else 'tensor-free (Gauss-Newton) variant'}",
                       "mode": a.mode, "horizon": T, "seeds_per_gpu": S, "n_alpha": a.n_alpha, "parallelism": f"seeds x{world}"},
            "roofline": {"kernel": "bwd_contract (K3: V_x-contracted f_xx, f_ux, f_uu)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(S),
                         "bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_s * 1e6, "launches": n_a},
            "phases_ms_per_step": {k: v / a.steps for k, v in phase_ms.items()},
            "kernels_ms_per_step": {"bwd_contract": ms_a / a.steps,
                                    "bwd_riccati_and_gaps": (phase_ms["backward"] - ms_a) / a.steps, "fwd_rollout": ms_f / a.steps,
                                    "lin_first": ms_l1 / a.steps, "lin_second": ms_l2 / a.steps},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, model)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()          # rank 0 may still be timing the CPU baseline: leave together
        dist.destroy_process_group()


def pmc_traffic(S):
    """HBM bytes per K3 launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 for gfx950's wide-read
    under-count + WRITE_SIZE, separate --pmc runs: profiles/k3_traffic.json), valid for the profiled batch only"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "k3_traffic.json")))
        return float(rec["bytes_per_launch"]) if int(rec["batch"]) == S else None
    except Exception:
        return None


def cpu_baseline(a, model):
    """The CPU oracle (a port of the reference algorithm; the reference itself cannot be built here) timed on
    this box's host cores, single thread like the reference (pinocchio_model.ipp:121), on a bounded sample:
    one instance, a shorter horizon, one full iteration; scaled linearly in the horizon to T (every phase is
    O(T))."""
    import tempfile
    from oracle import binding
    from oracle.binding import Oracle
    tmp = os.path.join(tempfile.gettempdir(), f"libddp_oracle_native_{os.getpid()}.so")
    try:
        path = binding.build(force=True, march="native", out=tmp)
    except Exception:
        path = None
    full = a.mode == "full"
    Ts = a.cpu_sample_steps or a.horizon
    o = Oracle(model, Ts, dt=0.01, c=1.0, fd_mode=2 if full else 0, lib_path=path)
    us = 0.1 * np.random.default_rng(0xDD9000 + 3000).normal(size=Ts * model.nv)
    xs = o.rollout(np.zeros(2 * model.nv), us)
    mults = o.alloc_affine(0)
    t0 = time.perf_counter()
    d = o.compute_derivatives(xs, us)
    t1 = time.perf_counter()
    bw = o.backward(d, xs, mults, reg=0.0, mu=1e2, trace=False, heap_like=True)
    t2 = time.perf_counter()
    o.forward(xs, us, mults, bw["fb"], bw["mu"])
    t3 = time.perf_counter()
    scale = a.horizon / Ts
    total = (t3 - t0) * scale
    return {"value": 1.0 / total, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"1 instance, horizon {Ts} of {a.horizon} (scaled x{scale:g}), one iteration: linearise "
                      f"{(t1 - t0):.2f}s backward {(t2 - t1):.3f}s forward {(t3 - t2):.3f}s; gcc -O3 -march=native, 1 thread",
            "cpu_seconds": t3 - t0}


if __name__ == "__main__":
    main()
