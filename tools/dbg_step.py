import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ddp_pinocchio_amd import capi
from problems import make
for name in ("chain6", "tree38"):
    B = 4
    model, spec, o = make(name, 1, batch=B)
    rng = np.random.default_rng(0)
    with capi.Context(spec, flags=capi.FLAG_NO_TENSORS) as ctx:
        xs = rng.normal(size=(B, 2 * 2 * model.nv)); us = rng.normal(size=(B, model.nv))
        ctx.upload("X", xs); ctx.upload("U", us)
        ctx.rollout()
        got = ctx.download("X")
        for b in range(B):
            ref = o.eval_f(xs[b, :2 * model.nv], us[b])
            err = np.abs(got[b, 2 * model.nv:] - ref)
            print(name, b, "max abs err", err.max(), "scale", np.abs(ref).max(), "argmax", err.argmax())
