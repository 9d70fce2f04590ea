import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from problems import make, initial_trajectory
from ddp_pinocchio_amd import capi
from synth import rel_err
from test_dynamics_parity import DERIV_SEQS
T, mu = 200, 1e3
for sig in (0.01,):
    model, spec, o = make("chain6_frame", T, fd_mode=0)
    from problems import held_trajectory
    x0, us, xs = held_trajectory(o, model, seed=5, u_sigma=sig)
    print("sigma", sig, "max|x|", np.abs(xs).max())
    rng = np.random.default_rng(5)
    mults = o.alloc_affine(o.Etot); mults["origin"][:] = xs[:T * o.nx]
    mults["jac"][:o.Etot * o.n] = rng.normal(size=o.Etot * o.n)
    with capi.Context(spec, flags=capi.FLAG_TRACE | capi.FLAG_NO_TENSORS) as ctx:
        ctx.upload("X", xs); ctx.upload("U", us)
        for k in ("origin","val","jac"):
            s="MULT_"+k.upper()
            if ctx.seq_size(s): ctx.upload(s, mults[k][:ctx.seq_size(s)])
        ctx.linearize()
        d = o.alloc_derivs()
        for key, seq in DERIV_SEQS.items():
            sz = ctx.seq_size(seq)
            if sz: d[key][:sz] = ctx.download(seq, 0, 1)[0]
        print("max fx", np.abs(d["fx"]).max())
        ref = o.backward(d, xs, mults, 0.0, mu)
        rc, reg, mu_out, restarts = ctx.backward(0.0, mu)
        Vx = ctx.download("VX_TRACE")[0].reshape(T, -1); Vr = ref["Vx"].reshape(T, -1)
        K = ctx.download("FB_JAC")[0].reshape(T, -1); Kr = ref["fb"]["jac"][:K.size].reshape(T, -1)
        for t in (199, 198, 197, 190, 150, 100, 50, 10, 0):
            print(t, "Vx err %.2e |Vx| %.2e   K err %.2e |K| %.2e" % (rel_err(Vx[t], Vr[t]), np.abs(Vr[t]).max(), rel_err(K[t], Kr[t]), np.abs(Kr[t]).max()))
