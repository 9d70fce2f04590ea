"""Independent numpy / mpmath restatement of the reference's backward sweep, cost and indexers.

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/ddp_oracle.c (different language, different
primitives: einsum, numpy Cholesky / mpmath matrices) so that the two restatements pin each other:
tests/test_oracle_pinning.py compares them on seeded inputs and tests/golden/ holds vectors made by
tests/golden/make_golden.py from this file.

Follows include/ddp/ddp_bwd.ipp:26-154, include/ddp/detail/tensor.hpp:141-198,
include/ddp/detail/mat_seq.hpp:61-73, include/ddp/indexer.hpp:152-493 (file:line relative to the
reference root).
"""
import numpy as np


# ---------------------------------------------------------------------------------------------
# indexers (indexer.hpp): only what the reference's own tests exercise
# ---------------------------------------------------------------------------------------------
class RegularIndexer:  # indexer.hpp:249-289
    def __init__(self, begin, end, rows, cols=1):
        assert begin < end
        self.begin, self.end, self._rows, self._cols = begin, end, rows, cols

    def index_begin(self): return self.begin
    def index_end(self): return self.end
    def rows(self, t): return self._rows
    def cols(self, t): return self._cols
    def max_rows(self): return self._rows
    def max_cols(self): return self._cols
    def stride(self, t): return self._rows * self._cols
    def required_memory(self): return self._rows * self._cols * (self.end - self.begin)


class PeriodicRowFilter:  # indexer.hpp:395-447
    def __init__(self, idx, period, first_offset):
        assert period > 0 and first_offset < period
        self.idx, self.period, self.first_offset = idx, period, first_offset

    def index_begin(self): return self.idx.index_begin()
    def index_end(self): return self.idx.index_end()
    def rows(self, t): return self.idx.rows(t) if (t - self.idx.index_begin()) % self.period == self.first_offset else 0
    def cols(self, t): return 1
    def max_rows(self): return self.idx.max_rows()
    def max_cols(self): return 1
    def stride(self, t): return self.rows(t) * self.cols(t)
    def required_memory(self): return sum(self.stride(t) for t in range(self.index_begin(), self.index_end()))


class RangeRowFilter:  # indexer.hpp:328-393
    def __init__(self, idx, range_begin, range_end):
        self.idx, self.rb, self.re = idx, range_begin, range_end

    def index_begin(self): return self.idx.index_begin()
    def index_end(self): return self.idx.index_end()
    def rows(self, t): return self.idx.rows(t) if self.rb <= t < self.re else 0
    def cols(self, t): return 1
    def max_rows(self): return self.idx.max_rows()
    def max_cols(self): return 1
    def stride(self, t): return self.rows(t) * self.cols(t)
    def required_memory(self):
        return sum(self.stride(t) for t in range(max(self.index_begin(), self.rb), min(self.index_end(), self.re)))


class ShiftTimeIdx:  # indexer.hpp:291-318
    def __init__(self, idx, dt):
        self.idx, self.dt = idx, dt

    def index_begin(self): return self.idx.index_begin() - self.dt
    def index_end(self): return self.idx.index_end() - self.dt
    def rows(self, t): return self.idx.rows(t + self.dt)
    def cols(self, t): return self.idx.cols(t + self.dt)
    def max_rows(self): return self.idx.max_rows()
    def max_cols(self): return self.idx.max_cols()
    def stride(self, t): return self.idx.stride(t + self.dt)
    def required_memory(self): return self.idx.required_memory()


class RowConcat:  # indexer.hpp:152-195
    def __init__(self, l, r):
        assert l.index_begin() == r.index_begin() and l.index_end() == r.index_end()
        self.l, self.r = l, r

    def index_begin(self): return self.l.index_begin()
    def index_end(self): return self.l.index_end()
    def rows(self, t): return self.l.rows(t) + self.r.rows(t)
    def cols(self, t): return self.l.cols(t)
    def max_rows(self): return self.l.max_rows() + self.r.max_rows()
    def max_cols(self): return self.l.max_cols()
    def stride(self, t): return self.l.stride(t) + self.r.stride(t)
    def required_memory(self): return self.l.required_memory() + self.r.required_memory()


class OuterProd:  # indexer.hpp:197-247
    def __init__(self, l, r):
        assert l.index_begin() == r.index_begin() and l.index_end() == r.index_end()
        self.l, self.r = l, r

    def index_begin(self): return self.l.index_begin()
    def index_end(self): return self.l.index_end()
    def rows(self, t): return self.l.rows(t)
    def cols(self, t): return self.r.rows(t)
    def max_rows(self): return self.l.max_rows()
    def max_cols(self): return self.r.max_rows()
    def stride(self, t): return self.rows(t) * self.cols(t)
    def required_memory(self): return sum(self.stride(t) for t in range(self.index_begin(), self.index_end()))


def block_offset(idx, t):
    """memory offset of block t: sum of the strides before it (indexer.hpp:60-75 iterator walk)"""
    return sum(idx.stride(s) for s in range(idx.index_begin(), t))


def mat_block(data, idx, t):
    """column-major view of block t of a flat sequence (detail/mat_seq.hpp:61-73)"""
    r, c = idx.rows(t), idx.cols(t)
    o = block_offset(idx, t)
    return data[o:o + r * c].reshape((r, c), order="F")


# ---------------------------------------------------------------------------------------------
# flat sequence helpers for the solver-shaped data
# ---------------------------------------------------------------------------------------------
def mat(flat, off, r, c):
    return np.asarray(flat[off:off + r * c]).reshape((r, c), order="F")


def tens(flat, off, O, L, R):
    """T[i, j, k] at off + i + j*O + k*O*L (detail/tensor.hpp:141-147)"""
    return np.asarray(flat[off:off + O * L * R]).reshape((O, L, R), order="F")


def backward_numpy(T, n, m, nx, ne, d, xs, mults, reg, mu, max_restarts=64):
    """ddp_bwd.ipp:26-154 with numpy primitives.  Returns dict(k, K, origin, Vx, Vxx, reg, mu, restarts)."""
    ne = np.asarray(ne, dtype=np.int64)
    Epre = np.concatenate([[0], np.cumsum(ne)])
    restarts = 0
    while True:
        Vxx = mat(d["lfxx"], 0, n, n).copy()
        Vx = np.asarray(d["lfx"][:n]).copy()
        ks = np.zeros((T, m)); Ks = np.zeros((T, m, n)); orig = np.zeros((T, nx))
        Vxs = np.zeros((T, n)); Vxxs = np.zeros((T, n, n))
        failed = False
        for t in range(T - 1, -1, -1):
            e, E = int(ne[t]), int(Epre[t])
            lx = np.asarray(d["lx"][t * n:(t + 1) * n]); lu = np.asarray(d["lu"][t * m:(t + 1) * m])
            lxx = mat(d["lxx"], t * n * n, n, n); lux = mat(d["lux"], t * m * n, m, n); luu = mat(d["luu"], t * m * m, m, m)
            fx = mat(d["fx"], t * n * n, n, n); fu = mat(d["fu"], t * n * m, n, m)
            fxx = tens(d["fxx"], t * n ** 3, n, n, n); fux = tens(d["fux"], t * n * m * n, n, m, n); fuu = tens(d["fuu"], t * n * m * m, n, m, m)
            eqv = np.asarray(d["eq_val"][E:E + e]); eqx = mat(d["eq_x"], E * n, e, n); equ = mat(d["eq_u"], E * m, e, m)
            eqxx = tens(d["eq_xx"], E * n * n, e, n, n); equx = tens(d["eq_ux"], E * m * n, e, m, n); equu = tens(d["eq_uu"], E * m * m, e, m, m)
            pe = np.asarray(mults["val"][E:E + e]); pex = mat(mults["jac"], E * n, e, n)
            tmp = pe + mu * eqv
            tmp2 = pex + mu * eqx
            Qx = lx + fx.T @ Vx + eqx.T @ tmp + pex.T @ eqv
            Qu = lu + fu.T @ Vx + equ.T @ tmp
            Qxx = lxx + (fx.T @ Vxx) @ fx + eqx.T @ tmp2 + pex.T @ eqx + np.einsum("i,ijk->jk", tmp, eqxx) + np.einsum("i,ijk->jk", Vx, fxx)
            Quu = luu + (fu.T @ Vxx) @ fu + (equ.T @ equ) * mu + np.einsum("i,ijk->jk", tmp, equu) + np.einsum("i,ijk->jk", Vx, fuu)
            Qux = lux + (fu.T @ Vxx) @ fx + equ.T @ tmp2 + np.einsum("i,ijk->jk", tmp, equx) + np.einsum("i,ijk->jk", Vx, fux)
            A = np.tril(Quu + reg * np.eye(m))
            A = A + np.tril(A, -1).T   # Eigen LLT<Lower> reads the lower triangle only
            ok = True
            try:
                Lc = np.linalg.cholesky(A)
                if not np.all(np.isfinite(Lc)):
                    ok = True  # NaN passes Eigen's x <= 0 test; nothing to do here
            except np.linalg.LinAlgError:
                ok = False
            if not ok:
                if reg < mu:
                    reg = mu
                mu *= 2
                reg *= 2
                failed = True
                break
            import scipy.linalg as sla
            k = -sla.cho_solve((Lc, True), Qu)
            K = -sla.cho_solve((Lc, True), Qux)
            orig[t] = xs[t * nx:(t + 1) * nx]
            ks[t], Ks[t] = k, K
            Vx = Qx + Qux.T @ k
            Vxx = Qxx + Qux.T @ K
            Vxs[t], Vxxs[t] = Vx, Vxx
        if not failed:
            break
        restarts += 1
        if restarts > max_restarts:
            raise RuntimeError("max restarts")
    return dict(k=ks, K=Ks, origin=orig, Vx=Vxs, Vxx=Vxxs, reg=reg, mu=mu, restarts=restarts)


def backward_mpmath(T, n, m, nx, ne, d, xs, mults, reg, mu, dps=50):
    """The same recursion in mpmath at `dps` digits (no restarts expected: raises if a pivot <= 0).
    Used to bound the rounding error of the double restatements (small sizes only: pure Python)."""
    import mpmath as mp
    mp.mp.dps = dps
    ne = [int(v) for v in ne]
    Epre = [0]
    for v in ne:
        Epre.append(Epre[-1] + v)

    def M(a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        return mp.matrix(a.tolist()) if a.size else mp.matrix(a.shape[0], a.shape[1])

    def contract(v, Tn, L, R):
        out = mp.matrix(L, R)
        O = Tn.shape[0]
        for j in range(L):
            for k in range(R):
                s = mp.mpf(0)
                for i in range(O):
                    s += v[i] * mp.mpf(float(Tn[i, j, k]))
                out[j, k] = s
        return out

    reg = mp.mpf(reg); mu = mp.mpf(mu)
    Vxx = M(mat(d["lfxx"], 0, n, n)); Vx = M(np.asarray(d["lfx"][:n]))
    ks = np.zeros((T, m)); Ks = np.zeros((T, m, n)); Vxs = np.zeros((T, n)); Vxxs = np.zeros((T, n, n))
    for t in range(T - 1, -1, -1):
        e, E = ne[t], Epre[t]
        lx = M(d["lx"][t * n:(t + 1) * n]); lu = M(d["lu"][t * m:(t + 1) * m])
        lxx = M(mat(d["lxx"], t * n * n, n, n)); lux = M(mat(d["lux"], t * m * n, m, n)); luu = M(mat(d["luu"], t * m * m, m, m))
        fx = M(mat(d["fx"], t * n * n, n, n)); fu = M(mat(d["fu"], t * n * m, n, m))
        fxx = tens(d["fxx"], t * n ** 3, n, n, n); fux = tens(d["fux"], t * n * m * n, n, m, n); fuu = tens(d["fuu"], t * n * m * m, n, m, m)
        Qx = lx + fx.T * Vx
        Qu = lu + fu.T * Vx
        Qxx = lxx + fx.T * Vxx * fx
        Quu = luu + fu.T * Vxx * fu
        Qux = lux + fu.T * Vxx * fx
        if e > 0:
            eqv = M(d["eq_val"][E:E + e]); eqx = M(mat(d["eq_x"], E * n, e, n)); equ = M(mat(d["eq_u"], E * m, e, m))
            pe = M(mults["val"][E:E + e]); pex = M(mat(mults["jac"], E * n, e, n))
            tmp = pe + mu * eqv
            tmp2 = pex + mu * eqx
            Qx += eqx.T * tmp + pex.T * eqv
            Qu += equ.T * tmp
            Qxx += eqx.T * tmp2 + pex.T * eqx + contract(tmp, tens(d["eq_xx"], E * n * n, e, n, n), n, n)
            Quu += (equ.T * equ) * mu + contract(tmp, tens(d["eq_uu"], E * m * m, e, m, m), m, m)
            Qux += equ.T * tmp2 + contract(tmp, tens(d["eq_ux"], E * m * n, e, m, n), m, n)
        Qxx += contract(Vx, fxx, n, n)
        Quu += contract(Vx, fuu, m, m)
        Qux += contract(Vx, fux, m, n)
        A = mp.matrix(m, m)
        for i in range(m):
            for j in range(m):
                A[i, j] = Quu[max(i, j), min(i, j)] + (reg if i == j else 0)
        mp.cholesky(A)          # raises unless A is positive definite
        Ainv = mp.inverse(A)
        k = -(Ainv * Qu)
        K = -(Ainv * Qux)
        Vx = Qx + Qux.T * k
        Vxx = Qxx + Qux.T * K
        ks[t] = [float(k[i]) for i in range(m)]
        Ks[t] = [[float(K[i, j]) for j in range(n)] for i in range(m)]
        Vxs[t] = [float(Vx[i]) for i in range(n)]
        Vxxs[t] = [[float(Vxx[i, j]) for j in range(n)] for i in range(n)]
    return dict(k=ks, K=Ks, Vx=Vxs, Vxx=Vxxs)


# ---------------------------------------------------------------------------------------------
# forward dynamics in mpmath (Featherstone RBDA Table 7.1, dense 6x6 Pluecker transforms): pins the analytic
# derivatives of oracle/ddp_oracle.c (orc_aba_derivatives) well below double precision -- central differences at
# 50 digits with a step of 1e-20 have a truncation error of ~1e-40
# ---------------------------------------------------------------------------------------------
def aba_mpmath(model, q, v, tau, dps=50):
    """qdd = ABA(q, v, tau) for a tree of 1-DoF joints; `model` has the attributes of capi.BuiltinModel"""
    import mpmath as mp
    mp.mp.dps = dps
    N = int(model.nv)
    f = mp.mpf

    def M(rows):
        return mp.matrix([[f(x) for x in r] for r in rows])

    def skew(a):
        return mp.matrix([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])

    def crm(vv):
        w, l = skew(vv[0:3]), skew(vv[3:6])
        out = mp.zeros(6, 6)
        out[0:3, 0:3] = w; out[3:6, 0:3] = l; out[3:6, 3:6] = w
        return out

    X, S, vel, c, IA, pA = [None] * N, [None] * N, [None] * N, [None] * N, [None] * N, [None] * N
    for i in range(N):
        a = mp.matrix([f(x) for x in model.axis[i]])
        Rp = M(model.Rp[i])
        pp = mp.matrix([f(x) for x in model.pp[i]])
        if int(model.jtype[i]) == 0:
            K = skew(a)
            RJ = mp.eye(3) + mp.sin(f(q[i])) * K + (1 - mp.cos(f(q[i]))) * (K * K)
            Rc, r = Rp * RJ, pp
            S[i] = mp.matrix([a[0], a[1], a[2], 0, 0, 0])
        else:
            Rc, r = Rp, pp + Rp * (a * f(q[i]))
            S[i] = mp.matrix([0, 0, 0, a[0], a[1], a[2]])
        E = Rc.T
        Xi = mp.zeros(6, 6)
        Xi[0:3, 0:3] = E; Xi[3:6, 3:6] = E; Xi[3:6, 0:3] = -(E * skew(r))
        X[i] = Xi
        vJ = S[i] * f(v[i])
        par = int(model.parent[i])
        vel[i] = (X[i] * vel[par] if par >= 0 else mp.zeros(6, 1)) + vJ
        c[i] = crm(vel[i]) * vJ
        com = mp.matrix([f(x) for x in model.com[i]])
        cx = skew(com)
        mass = f(model.mass_j[i])
        I6 = mp.zeros(6, 6)
        I6[0:3, 0:3] = M(model.Ic[i]) + mass * (cx * cx.T); I6[0:3, 3:6] = mass * cx; I6[3:6, 0:3] = mass * cx.T
        I6[3:6, 3:6] = mass * mp.eye(3)
        IA[i] = I6
        pA[i] = -(crm(vel[i]).T) * (I6 * vel[i])          # v x* (I v), crf(v) = -crm(v)^T
    U, D, u = [None] * N, [None] * N, [None] * N
    for i in range(N - 1, -1, -1):
        U[i] = IA[i] * S[i]
        D[i] = (S[i].T * U[i])[0]
        u[i] = f(tau[i]) - (S[i].T * pA[i])[0]
        par = int(model.parent[i])
        if par >= 0:
            Ia = IA[i] - U[i] * U[i].T / D[i]
            pa = pA[i] + Ia * c[i] + U[i] * (u[i] / D[i])
            IA[par] = IA[par] + X[i].T * Ia * X[i]
            pA[par] = pA[par] + X[i].T * pa
    g = [f(x) for x in model.gravity]
    a0 = mp.matrix([0, 0, 0, -g[0], -g[1], -g[2]])
    acc, qdd = [None] * N, [None] * N
    for i in range(N):
        par = int(model.parent[i])
        ap = X[i] * (acc[par] if par >= 0 else a0) + c[i]
        qdd[i] = (u[i] - (U[i].T * ap)[0]) / D[i]
        acc[i] = ap + S[i] * qdd[i]
    return qdd


def aba_derivatives_mpmath(model, q, v, tau, cols, dps=50, h="1e-20"):
    """columns `cols` (each (which, j), which in 'q', 'v', 't') of the partials of ABA by central differences in mpmath"""
    import mpmath as mp
    mp.mp.dps = dps
    hh = mp.mpf(h)
    out = {}
    for which, j in cols:
        args = {"q": [mp.mpf(float(x)) for x in q], "v": [mp.mpf(float(x)) for x in v], "t": [mp.mpf(float(x)) for x in tau]}
        args[which][j] += hh
        fp = aba_mpmath(model, args["q"], args["v"], args["t"], dps)
        args[which][j] -= 2 * hh
        fm = aba_mpmath(model, args["q"], args["v"], args["t"], dps)
        out[which, j] = np.array([float((a - b) / (2 * hh)) for a, b in zip(fp, fm)])
    return out
