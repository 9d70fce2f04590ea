#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

The reference cannot be built or run here and its own tests hold no numbers for the DDP sweep,
so these vectors come from the INDEPENDENT numpy restatement (oracle/np_oracle.py), cross-checked at creation
time against the C oracle (oracle/ddp_oracle.c) to 1e-11: two restatements written separately agreeing is what
pins them.  Inputs are regenerated from seeds by tests/synth.py (numpy's PCG64 streams are stable); the small
cases also store their inputs verbatim.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddp_pinocchio_amd import capi  # noqa: E402  (only for the dims of the dummy model handed to the oracle)
from oracle import np_oracle as npo  # noqa: E402
from oracle.binding import Oracle  # noqa: E402
from synth import rel_err, synth_sweep_inputs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (nv, T, ne, seed, reg, mu, indefinite_at, store_inputs)
    "sweep_pendulum_T50": (1, 50, [0] * 48 + [1, 0], 4101, 0.0, 10.0, None, True),
    "sweep_chain6_T10_e6": (6, 10, [6] * 10, 4102, 0.0, 10.0, None, True),
    "sweep_tree38_T4_tensors": (38, 4, [0] * 4, 4103, 0.0, 10.0, None, False),
    "sweep_restart_nv3_T6": (3, 6, [0] * 6, 4105, 0.0, 0.25, 3, True),
}


def main():
    for name, (nv, T, ne, seed, reg, mu, indef, store) in CASES.items():
        n, m, nx = 2 * nv, nv, 2 * nv
        d, xs, us, mults = synth_sweep_inputs(T, nv, ne, seed=seed, indefinite_at=indef)
        rn = npo.backward_numpy(T, n, m, nx, ne, d, xs, mults, reg, mu)
        model = capi.BuiltinModel(capi.BUILTIN_PENDULUM)
        model.nv = nv
        rc = Oracle(model, T, ne=ne).backward(d, xs, mults, reg, mu)
        k_c = rc["fb"]["val"][:T * m].reshape(T, m)
        assert rel_err(k_c, rn["k"]) < 1e-11 and rel_err(rc["Vx"].reshape(T, n), rn["Vx"]) < 1e-11, name
        assert rc["restarts"] == rn["restarts"] and rc["reg"] == rn["reg"] and rc["mu"] == rn["mu"], name
        out = dict(nv=nv, T=T, ne=np.asarray(ne), seed=seed, reg_in=reg, mu_in=mu, indefinite_at=-1 if indef is None else indef,
                   reg_out=rn["reg"], mu_out=rn["mu"], restarts=rn["restarts"],
                   k=rn["k"], Vx=rn["Vx"],
                   K_fro=np.array([np.linalg.norm(rn["K"][t]) for t in range(T)]),
                   Vxx_fro=np.array([np.linalg.norm(rn["Vxx"][t]) for t in range(T)]))
        if store:
            out.update(K=rn["K"], Vxx=rn["Vxx"], xs=xs, us=us,
                       **{f"d_{k}": v for k, v in d.items()}, **{f"mult_{k}": v for k, v in mults.items()})
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "restarts", rn["restarts"], os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
