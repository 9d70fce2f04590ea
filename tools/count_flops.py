#!/usr/bin/env python3
"""Counts the FP64 operations of one forward-dynamics evaluation of a tree model at the three levels of the mode-2
stencil (bench.py: FLOPS_FULL / FLOPS_VEL / FLOPS_TAU), by running the operation sequences of csrc/rbd.h on a counting
scalar: aba_tree (everything), aba_vu_cached (the configuration-dependent part E, r, U, 1/D, Ia is given) and aba_u_cached
(the (q, v)-dependent part cb, pA0, Ia cb is given as well).  An add, a subtract, a multiply and a divide count 1 each
(an FMA therefore 2); sincos counts 2.  Multiplications by the structural zeros / ones of a 1-DoF joint axis are counted as
written in rbd.h (the kernels do not special-case them either, except in the static-topology path).

    python tools/count_flops.py            # the Talos-like 38-joint tree
"""
import sys

import numpy as np

N_OPS = [0]


class C(float):
    """counting scalar (values are carried along so that the sequences can be checked against the oracle)"""
    __slots__ = ()

    def _w(self, v):
        N_OPS[0] += 1
        return C(v)

    def __add__(self, o): return self._w(float(self) + float(o))
    __radd__ = __add__
    def __sub__(self, o): return self._w(float(self) - float(o))
    def __rsub__(self, o): return self._w(float(o) - float(self))
    def __mul__(self, o): return self._w(float(self) * float(o))
    __rmul__ = __mul__
    def __truediv__(self, o): return self._w(float(self) / float(o))
    def __rtruediv__(self, o): return self._w(float(o) / float(self))
    def __neg__(self): return C(-float(self))


def cvec(a):
    return [C(v) for v in np.asarray(a, dtype=float).ravel()]


def cross3(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def mv3(A, x):
    return [A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2] for i in range(3)]


def mtv3(A, x):
    return [A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2] for i in range(3)]


def mm3(A, B):
    return [A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j] for i in range(3) for j in range(3)]


def mtm3(A, B):
    return [A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j] for i in range(3) for j in range(3)]


def sidx(r, c):
    return r * (r + 1) // 2 + c if r >= c else c * (c + 1) // 2 + r


def sym6_mv(I, x):
    out = []
    for r in range(6):
        s = C(0.0)
        for c in range(6):
            s = s + I[sidx(r, c)] * x[c]
        out.append(s)
    return out


def crm(v, m):
    a, b, c = cross3(v[:3], m[:3]), cross3(v[3:], m[:3]), cross3(v[:3], m[3:])
    return a + [b[k] + c[k] for k in range(3)]


def crf(v, f):
    a, b, c = cross3(v[:3], f[:3]), cross3(v[3:], f[3:]), cross3(v[:3], f[3:])
    return [a[k] + b[k] for k in range(3)] + c


def xform_motion(E, r, vp):
    t = cross3(r, vp[:3])
    u = [vp[3 + k] - t[k] for k in range(3)]
    return mv3(E, vp[:3]) + mv3(E, u)


def xform_force_T(E, r, fc):
    fp = mtv3(E, fc[:3]) + mtv3(E, fc[3:])
    t = cross3(r, fp[3:])
    return [fp[k] + t[k] for k in range(3)] + fp[3:]


def add_xtix(E, r, Ia, IAp):
    A = [Ia[sidx(i, j)] for i in range(3) for j in range(3)]
    B = [Ia[sidx(i, j + 3)] for i in range(3) for j in range(3)]
    Cc = [Ia[sidx(i + 3, j + 3)] for i in range(3) for j in range(3)]
    Ar, Br, Cr = mm3(mtm3(E, A), E), mm3(mtm3(E, B), E), mm3(mtm3(E, Cc), E)
    z = C(0.0)
    rx = [z, -r[2], r[1], r[2], z, -r[0], -r[1], r[0], z]
    rxC = mm3(rx, Cr)
    Bp = [Br[k] + rxC[k] for k in range(9)]
    BrT = [Br[3 * j + i] for i in range(3) for j in range(3)]
    rxBt, Brx, rxCrx = mm3(rx, BrT), mm3(Br, rx), mm3(rxC, rx)
    for i in range(3):
        for j in range(i + 1):
            IAp[sidx(i, j)] = IAp[sidx(i, j)] + (Ar[3 * i + j] + rxBt[3 * i + j] - Brx[3 * i + j] - rxCrx[3 * i + j])
            IAp[sidx(i + 3, j + 3)] = IAp[sidx(i + 3, j + 3)] + Cr[3 * i + j]
    for i in range(3):
        for j in range(3):
            IAp[sidx(j + 3, i)] = IAp[sidx(j + 3, i)] + Bp[3 * i + j]


def joint_placement(model, i, q):
    Rp, a = cvec(model["Rp"][i]), cvec(model["axis"][i])
    if model["jtype"][i] == 0:
        N_OPS[0] += 2                                   # sincos
        s, c = C(np.sin(q)), C(np.cos(q))
        z = C(0.0)
        K = [z, -a[2], a[1], a[2], z, -a[0], -a[1], a[0], z]
        K2 = mm3(K, K)
        omc = 1.0 - c
        RJ = [s * K[k] + omc * K2[k] for k in range(9)]
        RJ[0] = RJ[0] + 1.0; RJ[4] = RJ[4] + 1.0; RJ[8] = RJ[8] + 1.0
        Rc = mm3(Rp, RJ)
        E = [Rc[3 * l + k] for k in range(3) for l in range(3)]
        return E, cvec(model["pp"][i])
    d = [a[k] * q for k in range(3)]
    Rd = mv3(Rp, d)
    E = [Rp[3 * l + k] for k in range(3) for l in range(3)]
    return E, [C(model["pp"][i][k]) + Rd[k] for k in range(3)]


def aba(model, q, v, tau, level):
    """level 1: everything (aba_tree); 2: E, r, U, 1/D, Ia cached (aba_vu_cached); 3: cb, pA0, Ia cb cached too (aba_u_cached).
    The cached quantities are computed outside the count."""
    N = model["nv"]
    par, jt = model["parent"], model["jtype"]
    I6 = [cvec(model["I6"][i]) for i in range(N)]
    cnt_q = cnt_v = 0
    E, R, vel, cb, pA0 = [None] * N, [None] * N, [None] * N, [None] * N, [None] * N
    before = N_OPS[0]
    for i in range(N):
        E[i], R[i] = joint_placement(model, i, C(q[i]))
    cnt_q += N_OPS[0] - before
    before = N_OPS[0]
    for i in range(N):
        a = cvec(model["axis"][i]); o = 0 if jt[i] == 0 else 3
        vJ = [C(0.0)] * 6
        for k in range(3):
            vJ[o + k] = a[k] * C(v[i])
        vl = xform_motion(E[i], R[i], vel[par[i]]) if par[i] >= 0 else [C(0.0)] * 6
        vel[i] = [vl[k] + vJ[k] for k in range(6)]
        cb[i] = crm(vel[i], vJ)
        pA0[i] = crf(vel[i], sym6_mv(I6[i], vel[i]))
    cnt_v += N_OPS[0] - before
    IA = [list(I6[i]) for i in range(N)]
    pA = [list(pA0[i]) for i in range(N)]
    U, Dinv, uu, Iac = [None] * N, [None] * N, [None] * N, [None] * N
    for i in range(N - 1, -1, -1):
        a = cvec(model["axis"][i]); o = 0 if jt[i] == 0 else 3
        before = N_OPS[0]
        U[i] = [IA[i][sidx(r, o)] * a[0] + IA[i][sidx(r, o + 1)] * a[1] + IA[i][sidx(r, o + 2)] * a[2] for r in range(6)]
        d = a[0] * U[i][o] + a[1] * U[i][o + 1] + a[2] * U[i][o + 2]
        Dinv[i] = 1.0 / d
        Ia = None
        if par[i] >= 0:
            Ia = [C(0.0)] * 21
            for r in range(6):
                for c in range(r + 1):
                    Ia[sidx(r, c)] = IA[i][sidx(r, c)] - U[i][r] * U[i][c] * Dinv[i]
            add_xtix(E[i], R[i], Ia, IA[par[i]])
        cnt_q += N_OPS[0] - before
        before = N_OPS[0]
        if par[i] >= 0:
            Iac[i] = sym6_mv(Ia, cb[i])
        cnt_v += N_OPS[0] - before
        sp = a[0] * pA[i][o] + a[1] * pA[i][o + 1] + a[2] * pA[i][o + 2]
        uu[i] = C(tau[i]) - sp
        if par[i] >= 0:
            pa = [pA[i][k] + Iac[i][k] + U[i][k] * (uu[i] * Dinv[i]) for k in range(6)]
            fp = xform_force_T(E[i], R[i], pa)
            pA[par[i]] = [pA[par[i]][k] + fp[k] for k in range(6)]
    acc = [None] * N
    qdd = [None] * N
    g = model["gravity"]
    for i in range(N):
        a = cvec(model["axis"][i]); o = 0 if jt[i] == 0 else 3
        src = acc[par[i]] if par[i] >= 0 else [C(0.0)] * 3 + [C(-g[0]), C(-g[1]), C(-g[2])]
        ap = xform_motion(E[i], R[i], src)
        s = C(0.0)
        for k in range(6):
            ap[k] = ap[k] + cb[i][k]
            s = s + U[i][k] * ap[k]
        qdd[i] = (uu[i] - s) * Dinv[i]
        acc[i] = list(ap)
        for k in range(3):
            acc[i][o + k] = acc[i][o + k] + a[k] * qdd[i]
    return [float(x) for x in qdd], cnt_q, cnt_v


def main():
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
    from ddp_pinocchio_amd import capi
    from oracle.binding import Oracle
    bm = capi.BuiltinModel(capi.BUILTIN_TREE38, 1)
    o = Oracle(bm, 1)
    N = bm.nv
    I6 = []
    for i in range(N):
        c = bm.com[i]; m = bm.mass_j[i]
        cx = np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])
        M = np.zeros((6, 6))
        M[:3, :3] = bm.Ic[i] + m * cx @ cx.T; M[:3, 3:] = m * cx; M[3:, :3] = m * cx.T; M[3:, 3:] = m * np.eye(3)
        I6.append([M[r, c_] for r in range(6) for c_ in range(r + 1)])
    model = dict(nv=N, parent=[int(p) for p in bm.parent], jtype=[int(j) for j in bm.jtype], axis=bm.axis, Rp=bm.Rp.reshape(N, 9),
                 pp=bm.pp, I6=I6, gravity=[float(g) for g in bm.gravity])
    rng = np.random.default_rng(0)
    q, v, tau = rng.normal(size=N), rng.normal(size=N), rng.normal(size=N)
    N_OPS[0] = 0
    qdd, cnt_q, cnt_v = aba(model, q, v, tau, 1)
    total = N_OPS[0]
    ref = o.aba(q, v, tau)
    err = float(np.max(np.abs(np.array(qdd) - ref)) / np.max(np.abs(ref)))
    assert err < 1e-10, err                      # the counted sequence IS the forward dynamics
    print(f"joints {N}: full evaluation {total} ops  (q-dependent part {cnt_q}, (q,v)-dependent part {cnt_v}, tau-dependent rest {total - cnt_q - cnt_v})")
    print(f"FLOPS_FULL = {total}   FLOPS_VEL = {total - cnt_q}   FLOPS_TAU = {total - cnt_q - cnt_v}   (check vs oracle ABA: rel err {err:.1e})")


if __name__ == "__main__":
    main()
