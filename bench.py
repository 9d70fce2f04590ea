#!/usr/bin/env python3
"""bench.py -- DDP iterations/s on the Talos-like 38-DoF tree, horizon T = 200 (BASELINE.json metric).

One "step" = one full DDP iteration of every resident instance on this GPU:
    linearise (first-order jacobians + second-order tensors)  ->  backward sweep (with tensors)
    ->  forward sweep with 8 batched line-search steps  ->  swap trajectories (+ the reg rule of ddp.hpp:819-824)
with all inputs resident in HBM.  Independent instances (random control seeds) are sharded across ranks with no
data-path collective; the only exchange is the best-cost pick: the local argmin on the device, ONE 16-byte RCCL all-gather
and the argmin of the G pairs on the device again (ddp_hip_shard_pick, csrc/comm.cpp) -- the N = 1 line runs the same pick
without the collective, so the lines are comparable.  `python3 bench.py --gpus N` without a launcher starts its N ranks
itself (fresh child processes, before anything in the parent touches a GPU).

Workloads (BASELINE.json configs):
    default                --seeds-per-gpu 64   config 4 at N = 1 / weak scaling ("scaling": "weak")
    --total-seeds 64       config 4 as worded: 64 seeds in total over the N ranks ("scaling": "strong")
    --seeds-per-gpu 1      config 3: one instance x 8 line-search alphas
The default run also times, after the timed region and under `extra`: config 3 (one instance), and the CONSTRAINED problem
classes the headline's data never reach (its value function is identically zero: DESIGN.md 4d) -- `constrained_frame`
(config 5's inputs: 3-row frame constraint at T-2, N(0,1)-seeded multiplier jacobians, mu = 1e3) and `constrained_config`
(test/pinocchio_ddp.cpp's shape at the Talos size: e = 38 at every step) -- V != 0, K != 0, the line search halves.

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

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_VECTOR_TFLOPS = 78.6  # half the guide's 157.3 TFLOP/s FP32 vector figure (an FP64 FMA issues at half the FP32 rate)

# FP64 operations of one forward-dynamics evaluation of the 38-joint tree at each level of the mode-2 stencil, counted
# from the operation sequences of csrc/rbd.h (tools/count_flops.py): a full articulated-body
# evaluation, one that reuses the configuration-dependent part (velocity level), one that also reuses the (q, v) part
FLOPS_FULL, FLOPS_VEL, FLOPS_TAU = 41_545, 15_052, 5_476


def stencil_flops_per_bt(nv, fd_mode):
    """second-order stage, per (instance, t)"""
    tri = nv * (nv - 1) // 2
    if fd_mode == 2:      # problem.hpp:152-298: (q,q) pairs | (q,v) + (v,v) pairs | (x,u) + (u,u) pairs, + 3 nv diagonal points
        n_full, n_vel, n_tau = tri + nv, nv * nv + tri + nv, 2 * nv * nv + tri + nv
        return n_full * FLOPS_FULL + n_vel * FLOPS_VEL + n_tau * FLOPS_TAU
    return 0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seeds-per-gpu", type=int, default=64)
    ap.add_argument("--total-seeds", type=int, default=0,
                    help="strong scaling (BASELINE config 4 as worded): this many seeds in total, split over the ranks")
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--mode", choices=["full", "gn"], default="full",
                    help="full = with second-order tensors (the reference's algorithm); gn = tensor-free variant")
    ap.add_argument("--fd-mode", type=int, choices=[1, 2], default=2,
                    help="second-order tensors: 2 = second differences of f (problem.hpp:152-298, FD f_x, f_u: the north star); "
                         "1 = forward differences of the analytic jacobians (problem.hpp:67-150: what the reference's UR5 drivers use)")
    ap.add_argument("--n-alpha", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-3 (single instance) leg after the timed region")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="development: do not bracket the kernel launches with HIP events (no roofline / per-kernel times)")
    ap.add_argument("--cpu-sample-steps", type=int, default=0, help="horizon of the CPU sample (0 = auto: 50)")
    ap.add_argument("--cpu-iterations", type=int, default=5)
    return ap.parse_args()


def make_comm(capi, dist, torch, rank, world, device, red_dev):
    """The library's RCCL communicator (capi.Comm, csrc/comm.cpp); the 128-byte id travels over torch.distributed."""
    uid = capi.Comm.unique_id() if rank == 0 else bytes(128)
    t = torch.tensor(list(uid), dtype=torch.uint8, device=red_dev)
    dist.broadcast(t, src=0)
    return capi.Comm(bytes(t.cpu().tolist()), rank, world, device)


def make_instances(capi, a, model, seeds, device, T):
    """a context holding one instance per global seed index in `seeds`"""
    full = a.mode == "full"
    fd_mode = a.fd_mode if full else 0
    S = len(seeds)
    spec = capi.ProblemSpec(model, T, dt=0.01, c=1.0, batch=S, fd_mode=fd_mode, first_order_fd=0 if (full and a.fd_mode == 1) else 1)
    ctx = capi.Context(spec, device=device, flags=0 if full else capi.FLAG_NO_TENSORS)
    nx, m = 2 * model.nv, model.nv
    # synthetic inputs: x0 neutral, u_t ~ N(0, 0.1^2), seeded by the GLOBAL instance index
    us = np.stack([0.1 * np.random.default_rng(0xDD9000 + 3000 + g).normal(size=T * m) for g in seeds])
    ctx.upload("X", np.zeros((S, (T + 1) * nx)))
    ctx.upload("U", us)
    ctx.rollout()
    ctx.upload("X_NEW", ctx.download("X"))
    ctx.upload("U_NEW", us)
    return ctx


class Iterator:
    def __init__(self, ctx, S, n_alpha):
        self.ctx, self.S, self.n_alpha = ctx, S, n_alpha
        self.reg = np.zeros(S)
        self.mu = np.full(S, 1e2)
        self.phase_ms = {"linearize": 0.0, "backward": 0.0, "forward": 0.0}

    def step(self, timed):
        ctx = self.ctx
        t0 = time.perf_counter()
        ctx.linearize()
        t1 = time.perf_counter()
        rc, self.reg, self.mu, restarts = ctx.backward(self.reg, self.mu)
        t2 = time.perf_counter()
        rc, step, dcost = ctx.forward(self.mu, n_alpha=self.n_alpha)
        t3 = time.perf_counter()
        self.reg = np.where(step >= 0.5, np.where(self.reg / 2 < 1e-5, 0.0, self.reg / 2), self.reg)   # ddp.hpp:819-824
        ctx.swap_traj()                                      # ddp.hpp:826
        if timed:
            self.phase_ms["linearize"] += (t1 - t0) * 1e3
            self.phase_ms["backward"] += (t2 - t1) * 1e3
            self.phase_ms["forward"] += (t3 - t2) * 1e3


def spawn_ranks(a):
    """`python3 bench.py --gpus N` with no launcher around it: start the N ranks as fresh child processes -- the parent has
    not imported torch nor touched a GPU, and it never execs -- relay rank 0's JSON line and leave with the worst exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    import torch
    import torch.distributed as dist
    from ddp_pinocchio_amd import capi
    from ddp_pinocchio_amd import shard as shard_rule

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
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

    T = a.horizon
    strong = a.total_seeds > 0
    total = a.total_seeds if strong else a.seeds_per_gpu * world
    # instance s -> rank s mod G (shard.instances_of_rank): the same rule in both modes
    mine = shard_rule.instances_of_rank(total, rank, world)
    S = len(mine)
    assert S >= 1, "fewer seeds than ranks"
    full = a.mode == "full"
    model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
    nv = model.nv
    ctx = make_instances(capi, a, model, mine, local_rank, T)
    info = ctx.info()
    it = Iterator(ctx, S, a.n_alpha)
    # (rehearsal on one GPU, backend gloo: RCCL cannot place two ranks on one device -- the exchange goes through shard.best_of)
    comm = make_comm(capi, dist, torch, rank, world, local_rank, red_dev) if (world > 1 and backend == "nccl") else None
    picks = []

    def one_iteration(timed):
        it.step(timed)
        # the one exchange step: best-cost pick over all seeds of all ranks.  Local argmin on the device, one 16-byte
        # all-gather, argmin of the G pairs (ddp_hip_shard_pick); with one rank the same device work and no collective
        if comm is not None or world == 1:
            picks.append(ctx.shard_pick(comm))
        else:   # gloo rehearsal with every rank on one GPU: the host mirror of the same rule
            total_cost = ctx.download("COSTS_OLD").sum(axis=1)
            picks.append(shard_rule.pick(total_cost, rank, world, device=red_dev))

    # HIP events around every launch of the roofline kernel (K3) and of the few-launch kernels; K4's 200 launches per sweep
    # are left out (an event pair costs stream time): its figure below is the backward phase minus K3.  Switched on ahead of
    # the warm-up: the sweep's launch graph (with its event-record nodes) is captured there, not in the timed region
    if a.no_kernel_events:
        ctx.profile_enable(False)
    else:
        ctx.profile_enable(True, kernels=[capi.K_BWD_ASSEMBLE, capi.K_FWD_ROLLOUT, capi.K_LIN_FIRST, capi.K_LIN_SECOND])
    for _ in range(a.warmup):
        one_iteration(False)
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

    # the same K iterations once more with no event brackets at all: the sweep then runs as one hipGraph launch (a profiled
    # sweep issues its 600 launches one by one), i.e. the product path as a host application drives it.  Reported beside the
    # instrumented numbers, never instead of them.
    plain_ms = None
    if not a.no_kernel_events and not a.no_extra:
        keep = dict(it.phase_ms)
        it.phase_ms = {k: 0.0 for k in keep}
        ctx.synchronize()
        tp = time.perf_counter()
        for _ in range(a.steps):
            one_iteration(True)
        ctx.synchronize()
        plain_s = time.perf_counter() - tp
        plain_ms = {"ms_per_step": plain_s / a.steps * 1e3, "phases_ms_per_step": {k: v / a.steps for k, v in it.phase_ms.items()}}
        it.phase_ms = keep

    out = None
    ctx_stream_bytes = ctx.bwd_stream_bytes()
    if rank == 0:
        ms_a, n_a = ctx.profile_get(capi.K_BWD_ASSEMBLE)
        ms_f, n_f = ctx.profile_get(capi.K_FWD_ROLLOUT)
        ms_l1, n_l1 = ctx.profile_get(capi.K_LIN_FIRST)
        ms_l2, n_l2 = ctx.profile_get(capi.K_LIN_SECOND)
        steps = max(a.steps, 1)
        # algorithmic bytes of ONE bwd_contract (K3) launch = one timestep of every resident instance: the three
        # tensors read once (n^3 + n^2 m + n m^2 doubles) + V_x read + the contracted blocks added to the dense terms K5 left
        # in P (n^2 + m n + m^2 doubles read and written back)
        n_, m_ = 2 * nv, nv
        words = (n_ ** 3 + n_ * n_ * m_ + n_ * m_ * m_) + n_ + 2 * (n_ * n_ + m_ * n_ + m_ * m_)
        launches_per_sweep = max(n_a / steps / T, 1e-9)           # > 1 when the batch is swept in groups on several streams
        bytes_per_launch = 8.0 * words * S / launches_per_sweep if full else 0.0
        avg_s = (ms_a / max(n_a, 1)) * 1e-3
        achieved = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        # Bytes K3 physically streams.  This context's own mode-2 tensors have structure that is exact in floating point: they
        # are symmetric in their two input indices, and the configuration rows of every column are zeros but two entries
        # (q+ = q + dt v is affine) -- K3 reads the lower halves of the columns j >= c of slab c plus those entries
        # (ddp_hip_bwd_stream_bytes: 2.1 of the 6.15 MB per (instance, t) of SURVEY.md 8(d)'s formula), and what is not read
        # is not written by the stencil.  `achieved` / `frac` are the PHYSICAL stream (what the HBM roofline bounds);
        # `survey_formula_*` is SURVEY.md 8(d)'s dense byte count over the same launch time, kept for comparison across rounds.
        stream = ctx_stream_bytes
        small = 8.0 * (n_ + 2 * (n_ * n_ + m_ * n_ + m_ * m_))          # V_x + the read-modify-write of the dense terms
        bytes_read = (stream + small) * S / launches_per_sweep if full else 0.0
        survey_gbs = achieved
        achieved = bytes_read / avg_s / 1e9 if avg_s > 0 else 0.0
        # the sweep-level figure SURVEY.md 8(d) / BASELINE.md 3 define: B_bwd of every resident instance / the time of the
        # whole backward phase (K3 + K4 + launch gaps + the status read-back)
        t_bwd = it.phase_ms["backward"] / steps * 1e-3
        sweep_bytes = ctx.bwd_algorithmic_bytes() * S
        sweep_gbs = sweep_bytes / t_bwd / 1e9 if t_bwd > 0 else 0.0
        lin2_s = ms_l2 / steps * 1e-3
        lin2_flops = stencil_flops_per_bt(nv, a.fd_mode if full else 0) * S * T
        # tensor bytes the stencil writes: all of them, less the mirror images the symmetric sweep never reads (formed on demand)
        # (what the sweep never reads -- mirror images, zero configuration rows -- is not written either: the same byte count)
        lin2_bytes = float(stream) * S * T if full else 0.0
        if full and a.fd_mode == 1:                               # the analytic mode-1 pass writes every f_xx / f_ux slab whole (zero rows included)
            lin2_bytes = 8.0 * (n_ ** 3 + n_ * n_ * m_) * S * T
        out = {
            "metric": "DDP iterations/sec (fwd+bwd sweep), Talos nq=38 T=200",
            "value": total * a.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Talos-like 38-DoF tree (nq=nv=38, n=76, m=38), T={T}, "
                                   + (f"{total} seeds in total over {world} GPU(s)" if strong else f"{a.seeds_per_gpu} seeds/GPU")
                                   + f" x {a.n_alpha} line-search alphas, "
                                   + (("full DDP (FD f_x,f_u + FD f_xx,f_ux,f_uu mode 2)" if a.fd_mode == 2 else
                                       "full DDP (analytic ABA derivatives + FD tensors mode 1)") if full else "tensor-free (Gauss-Newton) variant"),
                       "mode": a.mode, "fd_mode": a.fd_mode if full else 0, "horizon": T, "seeds_per_gpu": S, "total_seeds": total,
                       "n_alpha": a.n_alpha, "parallelism": f"seeds x{world}", "paths": info},
            "roofline": {"kernel": "bwd_contract (K3: V_x-contracted f_xx, f_ux, f_uu)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(S / launches_per_sweep, a.fd_mode if full else 0),
                         "bytes_per_launch": bytes_read, "avg_launch_us": avg_s * 1e6, "launches": n_a,
                         "survey_formula_bytes_per_launch": bytes_per_launch, "survey_formula_gbs": survey_gbs,
                         "survey_formula_frac": survey_gbs / HBM_PEAK_GBS,
                         # whole backward phase: B_bwd x instances / t_backward (SURVEY.md 8d's definition)
                         "sweep_achieved": sweep_gbs, "sweep_frac": sweep_gbs / HBM_PEAK_GBS,
                         "sweep_bytes": sweep_bytes, "sweep_ms": t_bwd * 1e3},
            "lin_second": {"bound": "fp64-valu", "flops": lin2_flops, "achieved": lin2_flops / lin2_s / 1e12 if lin2_s > 0 else 0.0,
                           "peak": FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                           "frac": lin2_flops / lin2_s / 1e12 / FP64_VECTOR_TFLOPS if lin2_s > 0 else 0.0,
                           "bytes_written": lin2_bytes, "write_gbs": lin2_bytes / lin2_s / 1e9 if lin2_s > 0 else 0.0, "ms": lin2_s * 1e3},
            "phases_ms_per_step": {k: v / steps for k, v in it.phase_ms.items()},
            "uninstrumented": None if plain_ms is None else {
                **plain_ms, "iterations_per_s_this_rank": S / (plain_ms["ms_per_step"] * 1e-3),
                "sweep_frac": sweep_bytes / (plain_ms["phases_ms_per_step"]["backward"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "best_pick": {"collective": "one 16-byte ncclAllGather per iteration" if world > 1 else "none (one rank: device argmin only)",
                          "last": {"cost": picks[-1][0], "global_instance": picks[-1][1]} if picks else None},
            "kernels_ms_per_step": {"bwd_contract": ms_a / steps,
                                    "bwd_riccati_and_gaps": (it.phase_ms["backward"] - ms_a) / steps, "fwd_rollout": ms_f / steps,
                                    "lin_first": ms_l1 / steps, "lin_second": ms_l2 / steps},
        }
    ctx.close()
    if rank == 0:
        extra = {}
        if world == 1 and not a.no_extra and S != 1:
            extra["config3_single_instance"] = single_instance_leg(capi, a, model, local_rank, T)
            if full and a.fd_mode == 2:
                try:
                    extra["reference_drivers_mode1"] = mode1_leg(capi, a, model, local_rank, T)
                except Exception as exc:
                    extra["reference_drivers_mode1"] = {"error": repr(exc)}
                for key, name in (("constrained_frame", "tree38_frame"), ("constrained_config", "tree38_config"),
                                  ("config5_free_flyer_frame", "tree38ff_frame")):
                    try:
                        extra[key] = constrained_leg(capi, a, name, local_rank, T)
                    except Exception as exc:            # a failure here must not cost the headline line
                        extra[key] = {"error": repr(exc)}
        if extra:
            out["extra"] = extra
        if not a.no_cpu_baseline and a.cpu_iterations > 0:
            out["cpu_baseline"] = cpu_baseline(a, model)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()          # rank 0 may still be timing the CPU baseline: leave together
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


def single_instance_leg(capi, a, model, device, T):
    """BASELINE config 3 as worded: ONE instance x 8 line-search alphas on one GPU (latency-bound: the two sequential
    sweeps cost the same for 1 instance as for 64)"""
    ctx = make_instances(capi, a, model, [0], device, T)
    it = Iterator(ctx, 1, a.n_alpha)
    it.step(False)
    ctx.synchronize()
    k = 3
    t0 = time.perf_counter()
    for _ in range(k):
        it.step(True)
    ctx.synchronize()
    el = time.perf_counter() - t0
    ctx.close()
    return {"workload": f"1 instance x {a.n_alpha} alphas, T={T}", "iterations_per_s": k / el, "ms_per_iteration": el / k * 1e3,
            "phases_ms": {p: v / k for p, v in it.phase_ms.items()}}


def mode1_leg(capi, a, model, device, T):
    """The headline workload in the derivative mode of the reference's own drivers (test/pinocchio_ddp.cpp:60,
    dy{model, dt, false} + second_order_finite_diff mode 1): analytic f_x, f_u (computeABADerivatives restated) and their
    forward differences as tensors -- same seeds, same inputs, full DDP."""
    import copy
    b = copy.copy(a)
    b.fd_mode = 1
    S = a.seeds_per_gpu
    ctx = make_instances(capi, b, model, list(range(S)), device, T)
    it = Iterator(ctx, S, a.n_alpha)
    it.step(False)
    ctx.synchronize()
    k = 3
    t0 = time.perf_counter()
    for _ in range(k):
        it.step(True)
    ctx.synchronize()
    el = time.perf_counter() - t0
    info = ctx.info()
    ctx.close()
    return {"workload": f"{S} seeds x {a.n_alpha} alphas, T={T}, analytic first order + fd_mode 1", "iterations_per_s": S * k / el,
            "ms_per_step": el / k * 1e3, "phases_ms": {p: v / k for p, v in it.phase_ms.items()},
            "paths": {"lin_path": info["lin_path"], "first_order": info["first_order"], "bwd_path": info["bwd_path"], "fwd_path": info["fwd_path"]}}


def constrained_leg(capi, a, name, device, T, seeds=None, iters=2):
    """One of the constrained problem classes at the bench size, on the headline's own kind of inputs (x0 neutral,
    u ~ N(0, 0.1^2) per global seed): tree38_config -- test/pinocchio_ddp.cpp's shape, the neutral configuration asked for
    at every step (e = 38) -- or tree38_frame -- config 5's shape, a 3-row frame-position constraint at T-2; two time
    shifts, multipliers zero with N(0, 0.01^2)-seeded jacobians, mu = 1e3, reg = 0.
    Every iteration is the reference's real one: update_derivatives (compute_derivatives, update_origin x 2, optimality,
    multiplier update when its test passes: ddp.hpp:642-696) + backward_pass + forward_pass + swap (ddp.hpp:804-826).
    Full DDP is tried first (mode-2 tensors of f and of the constraint chain in the sweep).  At T = 200 in double Q_uu does
    not turn positive definite on any Talos-size constrained input tried (free rollouts, zero gravity, posture-holding
    trajectories: tools/probe_constrained.py; the CPU restatement of the reference's algorithm agrees, and the reference
    itself runs this shape in 500-digit mpfr: DESIGN.md 4d): then the tensors are still generated and timed, and the
    iterations run tensor-free (Gauss-Newton sweeps: V != 0, K != 0, LLT restarts and step halvings happen)."""
    from problems import make, neutral_state
    S = seeds or a.seeds_per_gpu
    model, spec, o = make(name, T, batch=S, fd_mode=2)
    m, nx, n_, Etot = o.m, o.nx, o.n, o.Etot
    us = np.stack([0.1 * np.random.default_rng(0xDD9000 + 5000 + g).normal(size=T * m) for g in range(S)])
    jac = np.stack([0.01 * np.random.default_rng(0xDD9000 + 6000 + g).normal(size=Etot * n_) for g in range(S)])
    state = {}

    def load(ctx):
        if "xs" not in state:
            x_init = np.zeros((S, (T + 1) * nx))
            x_init[:, :nx] = neutral_state(model)            # (a free-flyer root starts at the unit quaternion)
            ctx.upload("X", x_init); ctx.upload("U", us)
            ctx.rollout()
            state["xs"] = ctx.download("X")
        xs = state["xs"]
        ctx.upload("X", xs); ctx.upload("U", us); ctx.upload("X_NEW", xs); ctx.upload("U_NEW", us)
        ctx.upload("MULT_ORIGIN", np.ascontiguousarray(xs[:, :T * nx]))
        ctx.upload("MULT_VAL", np.zeros((S, Etot)))
        ctx.upload("MULT_JAC", jac)

    def iterate(ctx, max_restarts):
        mu = np.full(S, 1e3); reg = np.zeros(S); w = np.full(S, 1e-1); n = np.full(S, 10.0)
        ph = {"linearize": 0.0, "outer": 0.0, "backward": 0.0, "forward": 0.0}
        steps_seen, restarts_seen = [], 0
        ctx.linearize()                                      # ddp.hpp:768-772 ahead of the loop (also the warm-up)
        _, _r, mu, _ = ctx.backward(reg, mu, max_restarts)
        _, step, _ = ctx.forward(mu, n_alpha=a.n_alpha)
        ctx.synchronize()
        t_all = time.perf_counter()
        for _ in range(iters):
            t0 = time.perf_counter()
            ctx.linearize()
            t1 = time.perf_counter()
            ctx.update_origin(0); ctx.update_origin(1)
            oo, cc = ctx.optimality(mu)
            upd = (oo < w) & (cc < n)
            if upd.any():
                ctx.update_multipliers(np.where(upd, mu, 0.0))
                oo2, _ = ctx.optimality(mu)
                n = np.where(upd, oo2 / mu ** 0.1, n); w = np.where(upd, w / mu, w)
            mu = np.where((oo < w) & ~(cc < n), mu * 10, mu)
            t2 = time.perf_counter()
            _, reg, mu, restarts = ctx.backward(reg, mu, max_restarts)
            t3 = time.perf_counter()
            _, step, dcost = ctx.forward(mu, n_alpha=a.n_alpha)
            t4 = time.perf_counter()
            reg = np.where(step >= 0.5, np.where(reg / 2 < 1e-5, 0.0, reg / 2), reg)
            ctx.swap_traj()
            ph["linearize"] += (t1 - t0) * 1e3; ph["outer"] += (t2 - t1) * 1e3
            ph["backward"] += (t3 - t2) * 1e3; ph["forward"] += (t4 - t3) * 1e3
            steps_seen += [float(x) for x in step]
            restarts_seen += int(np.sum(restarts))
        ctx.synchronize()
        el = time.perf_counter() - t_all
        return {"iterations_per_s": S * iters / el, "ms_per_iteration": el / iters * 1e3,
                "phases_ms": {k: v / iters for k, v in ph.items()}, "paths": ctx.info(),
                "accepted_steps": {"min": min(steps_seen), "median": float(np.median(steps_seen)), "max": max(steps_seen)},
                "llt_restarts": restarts_seen, "max_abs_K_instance0": float(np.max(np.abs(ctx.download("FB_JAC", 0, 1))))}

    out = {"workload": f"{name}: Talos-like tree, T={T}, {S} seeds x {a.n_alpha} alphas, "
                       f"{'free-flyer root (nq=39), ' if 'ff' in name else ''}"
                       f"{'3-row frame constraint at T-2' if 'frame' in name else 'config constraint e=38 at every step'}, "
                       "two time shifts, N(0,0.01^2) multiplier jacobians, mu=1e3"}
    with capi.Context(spec, device=device) as ctx:
        load(ctx)
        try:
            out["full_ddp"] = iterate(ctx, 8)
        except capi.DdpHipError as exc:
            if exc.code != capi.E_MAX_RESTARTS:
                raise
            load(ctx)
            ctx.linearize(); ctx.synchronize()
            t0 = time.perf_counter()
            ctx.linearize(); ctx.synchronize()
            out["full_ddp"] = {"sweep": "Q_uu not positive definite within 8 restarts at this horizon in double (as on the CPU restatement of "
                                        "the reference's algorithm: DESIGN.md 4d); tensors generated and timed, the iterations below run tensor-free",
                               "linearize_ms": (time.perf_counter() - t0) * 1e3, "paths": ctx.info()}
    if "ff" not in name:
        # the same problem linearised in the reference drivers' derivative mode (analytic jacobians + their forward differences,
        # constraint tensors included; the reference asserts nq == nv in this mode, so not for the free-flyer shape)
        try:
            _, spec1, _ = make(name, T, batch=S, fd_mode=1, first_order_fd=0)
            with capi.Context(spec1, device=device) as ctx:
                load(ctx)
                ctx.linearize(); ctx.synchronize()
                t0 = time.perf_counter()
                ctx.linearize(); ctx.synchronize()
                out["reference_drivers_mode1"] = {"linearize_ms": (time.perf_counter() - t0) * 1e3, "paths": ctx.info()}
        except Exception as exc:
            out["reference_drivers_mode1"] = {"error": repr(exc)}
    if "iterations_per_s" not in out["full_ddp"]:
        _, spec0, _ = make(name, T, batch=S, fd_mode=0)
        with capi.Context(spec0, device=device, flags=capi.FLAG_NO_TENSORS) as ctx:
            load(ctx)
            try:
                out["tensor_free"] = iterate(ctx, 64)
            except capi.DdpHipError as exc:
                out["tensor_free"] = {"error": repr(exc)}
    return out


def pmc_traffic(S, fd_mode):
    """HBM bytes per K3 launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 for gfx950's wide-read
    under-count + WRITE_SIZE, separate --pmc runs: profiles/k3_traffic.json), valid for the profiled batch only"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "k3_traffic_mode1.json" if int(fd_mode) == 1 else "k3_traffic.json")))
        same = int(rec["batch"]) == int(round(S)) and int(rec.get("fd_mode", 2)) == int(fd_mode) and not os.environ.get("DDP_HIP_K3_NO_HALF")
        return float(rec["bytes_per_launch"]) if same else None
    except Exception:
        return None


def usable_cores():
    """host threads this process may really run on: the affinity mask, clipped by the cgroup CPU quota (a one-GPU share of a
    256-thread host is 16 threads; os.cpu_count() would report the whole host)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(a, model):
    """The CPU oracle (a port of the reference algorithm; the reference itself cannot be built here) timed on this box's
    host cores on a bounded sample of the same workload: one instance, a shorter horizon (every phase is O(T): scaled
    linearly to T), real consecutive DDP iterations (linearise -> backward -> forward -> swap), median over
    --cpu-iterations.  Legs: 1 thread "reference-like" (per-step heap temporaries as
    ddp_bwd.ipp:27-83; the headline `value`, since the reference is single-threaded: pinocchio_model.ipp:121), 1 thread
    "best-effort" (pre-allocated workspaces), and all host cores with one instance per thread."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding
    from oracle.binding import Oracle
    tmp = os.path.join(tempfile.gettempdir(), f"libddp_oracle_native_{os.getpid()}.so")
    try:
        path = binding.build(force=True, march="native", out=tmp)
    except Exception:
        path = None
    full = a.mode == "full"
    Ts = a.cpu_sample_steps or min(50, a.horizon)
    scale = a.horizon / Ts
    fd_mode = a.fd_mode if full else 0
    fo_fd = 0 if (full and a.fd_mode == 1) else 1

    def run(seed, iters, heap_like):
        o = Oracle(model, Ts, dt=0.01, c=1.0, fd_mode=fd_mode, first_order_fd=fo_fd, lib_path=path)
        us = 0.1 * np.random.default_rng(0xDD9000 + 3000 + seed).normal(size=Ts * model.nv)
        xs = o.rollout(np.zeros(2 * model.nv), us)
        mults = o.alloc_affine(0)
        reg, mu = 0.0, 1e2
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            d = o.compute_derivatives(xs, us)
            t1 = time.perf_counter()
            bw = o.backward(d, xs, mults, reg=reg, mu=mu, trace=False, heap_like=heap_like)
            t2 = time.perf_counter()
            step, xs_new, us_new, _ = o.forward(xs, us, mults, bw["fb"], bw["mu"])
            t3 = time.perf_counter()
            reg, mu = bw["reg"], bw["mu"]
            if step >= 0.5:
                reg = 0.0 if reg / 2 < 1e-5 else reg / 2
            xs, us = xs_new, us_new
            times.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
        return np.array(times)

    t_all0 = time.perf_counter()
    ref_like = run(0, a.cpu_iterations, True)
    best = run(0, max(2, a.cpu_iterations // 2), False)
    cores = usable_cores()
    tp0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:           # ctypes releases the GIL: one instance per thread
        list(ex.map(lambda s: run(s, 1, False), range(cores)))
    all_cores_s = time.perf_counter() - tp0
    med = np.median(ref_like, axis=0) * scale
    med_best = np.median(best, axis=0) * scale
    return {"value": 1.0 / med[0], "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"1 instance, horizon {Ts} of {a.horizon} (times scaled x{scale:g}), median of {a.cpu_iterations} consecutive "
                      f"iterations: linearise {med[1]:.2f}s backward {med[2]:.3f}s forward {med[3]:.3f}s; reference-like heap temporaries; "
                      f"gcc -O3 -march=native, 1 thread",
            "best_effort_1_thread": {"value": 1.0 / med_best[0], "sample": "pre-allocated workspaces, same sample"},
            "all_cores": {"value": cores / (all_cores_s * scale), "cores": cores,
                          "sample": f"{cores} instances, one per thread, one iteration each at horizon {Ts} (scaled)"},
            "cpu_seconds": time.perf_counter() - t_all0}


if __name__ == "__main__":
    main()
