"""development: the fused analytic kernel against the three-kernel form (DDP_HIP_ANA_SPLIT=1) -- same arithmetic, so bit for bit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ddp_pinocchio_amd import capi
from problems import held_trajectory, make

for name, T in (("tree38", 4), ("tree38_config", 3), ("tree38_frame", 4)):
    out = {}
    for split in (False, True):
        if split:
            os.environ["DDP_HIP_ANA_SPLIT"] = "1"
        else:
            os.environ.pop("DDP_HIP_ANA_SPLIT", None)
        model, spec, o = make(name, T, batch=2, fd_mode=1, first_order_fd=0)
        trajs = [held_trajectory(o, model, seed=5 + b, u_sigma=0.3) for b in range(2)]
        with capi.Context(spec) as ctx:
            ctx.upload("X", np.stack([tr[2] for tr in trajs])); ctx.upload("U", np.stack([tr[1] for tr in trajs]))
            ctx.linearize()
            out[split] = {k: ctx.download(k, 0, 2) for k in ("FX", "FU", "FXX", "FUX", "FUU", "EQ_X", "EQ_U", "EQ_XX", "EQ_UX", "EQ_UU") if ctx.seq_size(k)}
    for k in out[False]:
        same = np.array_equal(out[False][k], out[True][k])
        print(name, k, "bitwise" if same else ("max diff %g" % float(np.max(np.abs(out[False][k] - out[True][k])))))
