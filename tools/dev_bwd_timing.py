"""Development timing of the backward sweep at the Talos-like shape (random derivative inputs)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddp_pinocchio_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-tensors", action="store_true")
ap.add_argument("--no-events", action="store_true", help="no HIP events around the kernels: the sweep takes its hipGraph path")
a = ap.parse_args()

import ctypes as C
import torch

model = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
spec = capi.ProblemSpec(model, a.T, batch=a.batch)
flags = capi.FLAG_NO_TENSORS if a.no_tensors else 0
ctx = capi.Context(spec, flags=flags)
n, m, T, B = 76, 38, a.T, a.batch


def dev(name, shape):
    ptr = ctx.device_ptr(name)
    sz = ctx.seq_size(name) * B
    buf = (C.c_double * sz).from_address(0)  # placeholder, unused
    return ptr, sz


g = torch.Generator(device="cuda").manual_seed(0)


def view(name):
    """torch view over a resident sequence (plumbing only: random fill on the device)"""
    ptr, sz = ctx.device_ptr(name), ctx.seq_size(name) * B
    if sz == 0:
        return None
    # build a tensor from the raw pointer through the cuda array interface
    class _W:
        __cuda_array_interface__ = {"shape": (sz,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_W(), device="cuda")


def fill_normal(name, scale):
    v = view(name)
    if v is not None:
        v.normal_(0.0, scale, generator=g)


for name, sc in (("LFX", 0.1), ("LX", 0.1), ("LU", 0.1), ("LUX", 0.02), ("FU", 0.03), ("X", 1.0), ("U", 0.1)):
    fill_normal(name, sc)
if not a.no_tensors:
    for name in ("FXX", "FUX", "FUU"):
        fill_normal(name, 0.02 / n)
eye_n = torch.eye(n, device="cuda", dtype=torch.float64).reshape(-1)
eye_m = torch.eye(m, device="cuda", dtype=torch.float64).reshape(-1)
view("LFXX").view(B, n * n).copy_(0.3 * eye_n.expand(B, -1))
view("LXX").view(B * T, n * n).copy_(0.1 * eye_n.expand(B * T, -1))
view("LUU").view(B * T, m * m).copy_(1.0 * eye_m.expand(B * T, -1))
fx = view("FX").view(B * T, n * n)
fx.normal_(0.0, 0.05 / np.sqrt(n), generator=g)
fx.add_(eye_n)
torch.cuda.synchronize()

rc, reg, mu, rs = ctx.backward(0.0, 10.0)
print("warmup rc", rc, "restarts", rs.sum())
ctx.profile_enable(not a.no_events)
ctx.profile_reset()
if a.no_events:
    ctx.backward(0.0, 10.0)          # captures the graph
    ctx.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    ctx.backward(0.0, 10.0)
dt = (time.perf_counter() - t0) / a.reps
ms_a, n_a = ctx.profile_get(capi.K_BWD_ASSEMBLE)
ms_g, n_g = ctx.profile_get(capi.K_BWD_GAINS)
bytes_ = ctx.bwd_algorithmic_bytes() * B
print(f"batch {B} T {T}: sweep {dt * 1e3:.3f} ms  -> {B / dt:.1f} sweeps/s; algorithmic {bytes_ / 1e6:.1f} MB "
      f"-> {bytes_ / dt / 1e12:.3f} TB/s overall")
print(f"assemble: {ms_a / max(n_a, 1) * 1e3:.2f} us avg over {n_a} launches "
      f"({bytes_ / a.T / (ms_a / max(n_a, 1) * 1e-3) / 1e12:.3f} TB/s in-kernel)")
print(f"gains   : {ms_g / max(n_g, 1) * 1e3:.2f} us avg over {n_g} launches")
