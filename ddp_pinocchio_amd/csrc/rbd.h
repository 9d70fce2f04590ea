// rbd.h -- device rigid-body dynamics for trees of 1-DoF joints (articulated-body algorithm,
// Featherstone RBDA Table 7.1) and the closed-form pendulum.  Stands in for what the reference
// delegates to Pinocchio (pinocchio_model.ipp:353-355 aba, :222-321 Lie ops -- vector space here).
// One thread evaluates one forward dynamics; spatial vectors are [angular; linear].
#pragma once

#include "internal.h"
#include "lie.h"

namespace rbd {

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
// y = A x, y = A^T x  (row-major 3x3)
__device__ __forceinline__ void mv3(const double* A, const double* x, double* y) {
  y[0] = A[0] * x[0] + A[1] * x[1] + A[2] * x[2];
  y[1] = A[3] * x[0] + A[4] * x[1] + A[5] * x[2];
  y[2] = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
}
__device__ __forceinline__ void mtv3(const double* A, const double* x, double* y) {
  y[0] = A[0] * x[0] + A[3] * x[1] + A[6] * x[2];
  y[1] = A[1] * x[0] + A[4] * x[1] + A[7] * x[2];
  y[2] = A[2] * x[0] + A[5] * x[1] + A[8] * x[2];
}
__device__ __forceinline__ void mm3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// C = A^T B
__device__ __forceinline__ void mtm3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// E: rotation parent->child coordinates (row-major); r: child origin in parent coordinates
template <class M>
__device__ __forceinline__ void joint_placement(const M& m, int i, double q, double* E, double* r) {
  const double* Rp = m.Rp[i];
  const double* a = m.axis[i];
  if (m.jtype[i] == DDP_HIP_JOINT_REVOLUTE) {
    double s, c;
    sincos(q, &s, &c);
    const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
    double K2[9], RJ[9], Rc[9];
    mm3(K, K, K2);
    const double omc = 1.0 - c;
#pragma unroll
    for (int k = 0; k < 9; ++k) RJ[k] = s * K[k] + omc * K2[k];
    RJ[0] += 1.0; RJ[4] += 1.0; RJ[8] += 1.0;
    mm3(Rp, RJ, Rc);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) E[3 * k + l] = Rc[3 * l + k];
    r[0] = m.pp[i][0]; r[1] = m.pp[i][1]; r[2] = m.pp[i][2];
  } else {
    const double d[3] = {a[0] * q, a[1] * q, a[2] * q};
    double Rd[3];
    mv3(Rp, d, Rd);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) E[3 * k + l] = Rp[3 * l + k];
    r[0] = m.pp[i][0] + Rd[0]; r[1] = m.pp[i][1] + Rd[1]; r[2] = m.pp[i][2] + Rd[2];
  }
}

// motion transform parent -> child:  w_c = E w_p ;  v_c = E (v_p - r x w_p)
__device__ __forceinline__ void xform_motion(const double* E, const double* r, const double* vp, double* vc) {
  double t[3], u[3];
  cross3(r, vp, t);
  u[0] = vp[3] - t[0]; u[1] = vp[4] - t[1]; u[2] = vp[5] - t[2];
  mv3(E, vp, vc);
  mv3(E, u, vc + 3);
}
// force transform child -> parent (X^T):  f_p = E^T f_c ;  n_p = E^T n_c + r x f_p
__device__ __forceinline__ void xform_force_T(const double* E, const double* r, const double* fc, double* fp) {
  double t[3];
  mtv3(E, fc, fp);
  mtv3(E, fc + 3, fp + 3);
  cross3(r, fp + 3, t);
  fp[0] += t[0]; fp[1] += t[1]; fp[2] += t[2];
}

// symmetric 6x6, packed lower triangle: (r, c), c <= r, at r(r+1)/2 + c
__device__ __forceinline__ int sidx(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }

__device__ __forceinline__ void sym6_mv(const double* I, const double* x, double* y) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += I[sidx(r, c)] * x[c];
    y[r] = s;
  }
}

// v x m (motion),  v x* f (force)
__device__ __forceinline__ void crm(const double* v, const double* mm, double* out) {
  double a[3], b[3], c[3];
  cross3(v, mm, a);
  cross3(v + 3, mm, b);
  cross3(v, mm + 3, c);
  out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
  out[3] = b[0] + c[0]; out[4] = b[1] + c[1]; out[5] = b[2] + c[2];
}
__device__ __forceinline__ void crf(const double* v, const double* f, double* out) {
  double a[3], b[3], c[3];
  cross3(v, f, a);
  cross3(v + 3, f + 3, b);
  cross3(v, f + 3, c);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2];
  out[3] = c[0]; out[4] = c[1]; out[5] = c[2];
}

// IAp (packed) += X^T Ia X  with X = [E 0; -E rx E].
// Blocks of Ia (child coords): A = ang-ang, B = ang-lin, C = lin-lin.
//   A' = E^T A E, B' = E^T B E, C' = E^T C E;  C_p = C';  B_p = B' + rx C';
//   A_p = A' + rx B'^T - B' rx - rx C' rx
__device__ __forceinline__ void add_xtix(const double* E, const double* r, const double* Ia, double* IAp) {
  double A[9], B[9], Cc[9], T[9], Ar[9], Br[9], Cr[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[3 * i + j] = Ia[sidx(i, j)];
      B[3 * i + j] = Ia[sidx(i, j + 3)];   // row i (angular), column j+3 (linear)
      Cc[3 * i + j] = Ia[sidx(i + 3, j + 3)];
    }
  mtm3(E, A, T); mm3(T, E, Ar);
  mtm3(E, B, T); mm3(T, E, Br);
  mtm3(E, Cc, T); mm3(T, E, Cr);
  const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  double rxC[9], Bp[9], rxBt[9], Brx[9], rxCrx[9], BrT[9];
  mm3(rx, Cr, rxC);
#pragma unroll
  for (int k = 0; k < 9; ++k) Bp[k] = Br[k] + rxC[k];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) BrT[3 * i + j] = Br[3 * j + i];
  mm3(rx, BrT, rxBt);
  mm3(Br, rx, Brx);
  mm3(rxC, rx, rxCrx);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      IAp[sidx(i, j)] += Ar[3 * i + j] + rxBt[3 * i + j] - Brx[3 * i + j] - rxCrx[3 * i + j];
      IAp[sidx(i + 3, j + 3)] += Cr[3 * i + j];
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) IAp[sidx(j + 3, i)] += Bp[3 * i + j];   // (ang i, lin j) stored at (row j+3, col i)
}

// pendulum_model.hpp:105-114, g = 9.81 (:26)
__device__ __forceinline__ double pendulum_acc(const DevModel& m, double q, double tau) {
  return -9.81 / m.length * sin(q) + tau / m.mass;
}

// Articulated-body algorithm.  NJ = compile-time bound on the joint count (sizes the per-thread state).
// Per-joint private state: E | r | cb | pA0 | U | 1/D | u (32 doubles); the articulated inertias and the running
// vectors of the three tree passes live in a few slots (DevModel::slot_up / slot_down), allocated on the host so
// that every sum is formed in the order of the textbook loop (a parent's accumulator starts from its own value when
// its largest-index child contributes).
// x = A^-1 b for a symmetric positive definite 6 x 6 in packed lower-triangle storage (Cholesky): the 6-DoF root joint
__device__ __forceinline__ void sym6_solve(const double* A, const double* b, double* x) {
  double L[21];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double d = A[sidx(k, k)];
#pragma unroll
    for (int j = 0; j < k; ++j) d -= L[sidx(k, j)] * L[sidx(k, j)];
    d = sqrt(d);
    L[sidx(k, k)] = d;
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      double t = A[sidx(i, k)];
#pragma unroll
      for (int j = 0; j < k; ++j) t -= L[sidx(i, j)] * L[sidx(k, j)];
      L[sidx(i, k)] = t / d;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double t = b[i];
#pragma unroll
    for (int j = 0; j < i; ++j) t -= L[sidx(i, j)] * x[j];
    x[i] = t / L[sidx(i, i)];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double t = x[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) t -= L[sidx(j, i)] * x[j];
    x[i] = t / L[sidx(i, i)];
  }
}

// placement of joint i from the whole configuration vector: a free-flyer root reads (p, quaternion), the others one scalar
__device__ __forceinline__ void place(const DevModel& m, int i, const double* q, double* E, double* r) {
  if (i == 0 && m.ff) {
    double R[9];
    lie::quat_to_R(q + 3, R);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) E[3 * k + l] = R[3 * l + k];
    r[0] = q[0]; r[1] = q[1]; r[2] = q[2];
    return;
  }
  joint_placement(m, i, q[m.ff ? i + 6 : i], E, r);
}

template <int NJ>
__device__ void aba_tree(const DevModel& m, const double* q, const double* v, const double* tau, double* qdd) {
  const int N = m.nj;
  const int ff = m.ff, vo = ff ? 5 : 0;           // joint i >= 1 of a free-flyer model uses v[i + 5] (and q[i + 6])
  double E[NJ][9], R[NJ][3], cb[NJ][6], pA0[NJ][6], U[NJ][6], Dinv[NJ], uu[NJ];
  double slot[8][6], islot[8][21];
  double IAr[21], pAr[6];                         // articulated inertia / bias force of a 6-DoF root
  for (int i = 0; i < N; ++i) {
    place(m, i, q, E[i], R[i]);
    const double* a = m.axis[i];
    double vJ[6] = {0, 0, 0, 0, 0, 0}, vel[6];
    if (i == 0 && ff) {                           // S = identity on the body twist; v = [linear; angular]
      vJ[0] = v[3]; vJ[1] = v[4]; vJ[2] = v[5]; vJ[3] = v[0]; vJ[4] = v[1]; vJ[5] = v[2];
    } else {
      const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
      vJ[o] = a[0] * v[i + vo]; vJ[o + 1] = a[1] * v[i + vo]; vJ[o + 2] = a[2] * v[i + vo];
    }
    const int par = m.parent[i];
    if (par >= 0) xform_motion(E[i], R[i], slot[m.slot_down[par]], vel);
    else { for (int k = 0; k < 6; ++k) vel[k] = 0.0; }
    for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
    if (m.has_child[i]) { double* sl = slot[m.slot_down[i]]; for (int k = 0; k < 6; ++k) sl[k] = vel[k]; }
    crm(vel, vJ, cb[i]);
    double Iv[6];
    sym6_mv(m.I6[i], vel, Iv);
    crf(vel, Iv, pA0[i]);
  }
  for (int i = N - 1; i >= 0; --i) {
    const double* a = m.axis[i];
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    double IA[21], pAi[6];
    if (m.has_child[i]) {
      const int su = m.slot_up[i];
      for (int k = 0; k < 21; ++k) IA[k] = islot[su][k];
      for (int k = 0; k < 6; ++k) pAi[k] = slot[su][k];
    } else {
      for (int k = 0; k < 21; ++k) IA[k] = m.I6[i][k];
      for (int k = 0; k < 6; ++k) pAi[k] = pA0[i][k];
    }
    if (i == 0 && ff) {                           // resolved in the last pass
      for (int k = 0; k < 21; ++k) IAr[k] = IA[k];
      for (int k = 0; k < 6; ++k) pAr[k] = pAi[k];
      break;
    }
    double d = 0, sp = 0;
    for (int r = 0; r < 6; ++r) U[i][r] = IA[sidx(r, o)] * a[0] + IA[sidx(r, o + 1)] * a[1] + IA[sidx(r, o + 2)] * a[2];
    for (int k = 0; k < 3; ++k) { d += a[k] * U[i][o + k]; sp += a[k] * pAi[o + k]; }
    Dinv[i] = 1.0 / d;
    uu[i] = tau[i + vo] - sp;
    const int par = m.parent[i];
    if (par >= 0) {
      double Ia[21], pa[6], Iac[6], fp[6];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c <= r; ++c) Ia[sidx(r, c)] = IA[sidx(r, c)] - U[i][r] * U[i][c] * Dinv[i];
      sym6_mv(Ia, cb[i], Iac);
      for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[i][k] * (uu[i] * Dinv[i]);
      const int sp_ = m.slot_up[par];
      if (m.first_contrib[i]) {
        for (int k = 0; k < 21; ++k) islot[sp_][k] = m.I6[par][k];
        for (int k = 0; k < 6; ++k) slot[sp_][k] = pA0[par][k];
      }
      add_xtix(E[i], R[i], Ia, islot[sp_]);
      xform_force_T(E[i], R[i], pa, fp);
      for (int k = 0; k < 6; ++k) slot[sp_][k] += fp[k];
    }
  }
  for (int i = 0; i < N; ++i) {
    double ap[6];
    const int par = m.parent[i];
    if (par >= 0) xform_motion(E[i], R[i], slot[m.slot_down[par]], ap);
    else {
      const double a0[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
      xform_motion(E[i], R[i], a0, ap);
    }
    if (i == 0 && ff) {
      // S = I: qdd_s = IA^-1 (tau_s - pA) - a' in the spatial ordering [angular; linear]; tau / qdd are ordered [linear; angular]
      double rhs[6], qs[6];
      for (int k = 0; k < 3; ++k) { rhs[k] = tau[3 + k] - pAr[k]; rhs[3 + k] = tau[k] - pAr[3 + k]; }
      sym6_solve(IAr, rhs, qs);
      for (int k = 0; k < 6; ++k) { ap[k] += cb[0][k]; qs[k] -= ap[k]; }
      for (int k = 0; k < 3; ++k) { qdd[k] = qs[3 + k]; qdd[3 + k] = qs[k]; }
      if (m.has_child[0]) { double* sl = slot[m.slot_down[0]]; for (int k = 0; k < 6; ++k) sl[k] = ap[k] + qs[k]; }
      continue;
    }
    double s = 0;
    for (int k = 0; k < 6; ++k) { ap[k] += cb[i][k]; s += U[i][k] * ap[k]; }
    const double qd = (uu[i] - s) * Dinv[i];
    qdd[i + vo] = qd;
    if (m.has_child[i]) {
      const double* a = m.axis[i];
      const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
      double* sl = slot[m.slot_down[i]];
      for (int k = 0; k < 6; ++k) sl[k] = ap[k];
      sl[o] += a[0] * qd; sl[o + 1] += a[1] * qd; sl[o + 2] += a[2] * qd;
    }
  }
}

// dynamics_t::eval_to, problem.hpp:441-461:  q+ = q + dt v ;  v+ = v + dt * aba(q, v, u)
template <int NJ>
__device__ void eval_f(const DevModel& m, const double* x, const double* u, double* x_out) {
  const int nv = m.nv;
  if (m.kind == DDP_HIP_MODEL_PENDULUM) {
    const double acc = pendulum_acc(m, x[0], u[0]);
    const double vo = m.dt * x[1];
    x_out[0] = x[0] + vo;
    x_out[1] = x[1] + acc * m.dt;
    return;
  }
  double acc[NJ];
  if (m.ff) {
    // q+ = q (+) dt v on SE(3) x R^(nv-6) (model.integrate, problem.hpp:452); x = [q(nv+1); v(nv)]
    const int nq = nv + 1;
    aba_tree<NJ>(m, x, x + nq, u, acc);
    double dq[6];
    for (int k = 0; k < 6; ++k) dq[k] = m.dt * x[nq + k];
    lie::se3_integrate(x, dq, x_out);
    for (int i = 6; i < nv; ++i) { const double vo = m.dt * x[nq + i]; x_out[i + 1] = x[i + 1] + vo; }
    for (int i = 0; i < nv; ++i) x_out[nq + i] = x[nq + i] + acc[i] * m.dt;
    return;
  }
  aba_tree<NJ>(m, x, x + nv, u, acc);
  for (int i = 0; i < nv; ++i) {
    const double vo = m.dt * x[nv + i];
    x_out[i] = x[i] + vo;
    x_out[nv + i] = x[nv + i] + acc[i] * m.dt;
  }
}

// The configuration half of eval_f alone: q+ = q (+) dt v, the very operations of eval_f (problem.hpp:450-452) without the
// forward dynamics.  Both constraint kinds read q only (config_constraint_t problem.hpp:792-806, spatial_constraint_t
// :679-689), so the LAST look-ahead step of a constraint chain (constraint_advance_time_t::eval_to, :563-567) does not need
// its accelerations: bit-identical configurations at half the dynamics evaluations.  x_out's velocity half is left untouched.
template <int NJ>
__device__ __forceinline__ void eval_f_q(const DevModel& m, const double* x, double* x_out) {
  const int nv = m.nv;
  if (m.kind == DDP_HIP_MODEL_PENDULUM) { const double vo = m.dt * x[1]; x_out[0] = x[0] + vo; return; }
  if (m.ff) {
    const int nq = nv + 1;
    double dq[6];
    for (int k = 0; k < 6; ++k) dq[k] = m.dt * x[nq + k];
    lie::se3_integrate(x, dq, x_out);
    for (int i = 6; i < nv; ++i) { const double vo = m.dt * x[nq + i]; x_out[i + 1] = x[i + 1] + vo; }
    return;
  }
  for (int i = 0; i < nv; ++i) { const double vo = m.dt * x[nv + i]; x_out[i] = x[i] + vo; }
}

// world position of a frame fixed at `off` in joint `joint`'s frame (pinocchio_model.ipp:418-430), and
// optionally the reference's WORLD-frame jacobian rows (pinocchio_model.ipp:433-462): J is 3 x nv, ld 3
template <int NJ>
__device__ void frame_position(const DevModel& m, const double* q, double* p3, double* J) {
  // walk root -> joint along the ancestor chain
  int chain[NJ];
  int len = 0;
  for (int j = m.frame_joint; j >= 0; j = m.parent[j]) chain[len++] = j;
  double oR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, op[3] = {0, 0, 0};
  if (J) for (int k = 0; k < 3 * m.nv; ++k) J[k] = 0.0;
  for (int c = len - 1; c >= 0; --c) {
    const int i = chain[c];
    double E[9], r[3], Rc[9], t[3], nR[9];
    place(m, i, q, E, r);
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) Rc[3 * k + l] = E[3 * l + k];
    mv3(oR, r, t);
    op[0] += t[0]; op[1] += t[1]; op[2] += t[2];
    mm3(oR, Rc, nR);
    for (int k = 0; k < 9; ++k) oR[k] = nR[k];
    if (J) {
      const double lever[3] = {-op[0], -op[1], -op[2]};     // WORLD frame: lever arm to the world origin
      if (i == 0 && m.ff) {
        // free flyer: the columns of oMi.act(S), S = identity on [linear; angular] body twists
        for (int cc = 0; cc < 3; ++cc) {
          const double e[3] = {cc == 0 ? 1.0 : 0.0, cc == 1 ? 1.0 : 0.0, cc == 2 ? 1.0 : 0.0};
          double aw[3];
          mv3(oR, e, aw);
          J[3 * cc] = aw[0]; J[3 * cc + 1] = aw[1]; J[3 * cc + 2] = aw[2];
          cross3(aw, lever, J + 3 * (3 + cc));
        }
      } else {
        const int vi = m.ff ? i + 5 : i;
        double aw[3];
        mv3(oR, m.axis[i], aw);
        if (m.jtype[i] == DDP_HIP_JOINT_REVOLUTE) cross3(aw, lever, J + 3 * vi);
        else { J[3 * vi] = aw[0]; J[3 * vi + 1] = aw[1]; J[3 * vi + 2] = aw[2]; }
      }
    }
  }
  double t[3];
  mv3(oR, m.frame_off, t);
  p3[0] = op[0] + t[0]; p3[1] = op[1] + t[1]; p3[2] = op[2] + t[2];
}

}  // namespace rbd

// ---- articulated-body algorithm split by what each part depends on --------------------------------------------
// The finite-difference stencils evaluate the dynamics at thousands of points that share their configuration q:
// everything the ABA derives from q alone (joint placements, articulated inertias, U = IA S, 1/D, the projected
// inertia Ia) is computed once per distinct q (aba_qpart) and read back by every evaluation that only varies v or
// tau (aba_vu_cached).  The cached part is the very same sequence of operations as in aba_tree, so the split changes
// nothing but the amount of repeated work.
namespace rbd {

constexpr int QC_STRIDE = 40;   // per joint: E[9] r[3] U[6] Dinv Ia[21]

template <int NJ>
__device__ void aba_qpart(const DevModel& m, const double* q, double* __restrict__ qc) {
  const int N = m.nv;
  double E[NJ][9], R[NJ][3], IA[NJ][21];
  for (int i = 0; i < N; ++i) {
    joint_placement(m, i, q[i], E[i], R[i]);
    for (int k = 0; k < 21; ++k) IA[i][k] = m.I6[i][k];
  }
  for (int i = N - 1; i >= 0; --i) {
    const double* a = m.axis[i];
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    double U[6], d = 0;
    for (int r = 0; r < 6; ++r) U[r] = IA[i][sidx(r, o)] * a[0] + IA[i][sidx(r, o + 1)] * a[1] + IA[i][sidx(r, o + 2)] * a[2];
    for (int k = 0; k < 3; ++k) d += a[k] * U[o + k];
    const double dinv = 1.0 / d;
    double Ia[21];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c <= r; ++c) Ia[sidx(r, c)] = IA[i][sidx(r, c)] - U[r] * U[c] * dinv;
    const int par = m.parent[i];
    if (par >= 0) add_xtix(E[i], R[i], Ia, IA[par]);
    double* o_ = qc + i * QC_STRIDE;
    for (int k = 0; k < 9; ++k) o_[k] = E[i][k];
    for (int k = 0; k < 3; ++k) o_[9 + k] = R[i][k];
    for (int k = 0; k < 6; ++k) o_[12 + k] = U[k];
    o_[18] = dinv;
    for (int k = 0; k < 21; ++k) o_[19 + k] = Ia[k];
  }
}

template <int NJ>
__device__ void aba_vu_cached(const DevModel& m, const double* __restrict__ qc, const double* v, const double* tau, double* qdd) {
  // per-joint private state: cb | pA0 (12 doubles); the running sums of the tree passes live in a few slots
  const int N = m.nv;
  double cbp[NJ][12], uu[NJ], slot[8][6];
  for (int i = 0; i < N; ++i) {
    const double* E = qc + i * QC_STRIDE;
    const double* r = E + 9;
    const double* a = m.axis[i];
    double vJ[6] = {0, 0, 0, 0, 0, 0}, vel[6];
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    vJ[o] = a[0] * v[i]; vJ[o + 1] = a[1] * v[i]; vJ[o + 2] = a[2] * v[i];
    const int par = m.parent[i];
    if (par >= 0) xform_motion(E, r, slot[m.slot_down[par]], vel);
    else { for (int k = 0; k < 6; ++k) vel[k] = 0.0; }
    for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
    if (m.has_child[i]) { double* sl = slot[m.slot_down[i]]; for (int k = 0; k < 6; ++k) sl[k] = vel[k]; }
    crm(vel, vJ, cbp[i]);
    double Iv[6];
    sym6_mv(m.I6[i], vel, Iv);
    crf(vel, Iv, cbp[i] + 6);
  }
  for (int i = N - 1; i >= 0; --i) {
    const double* E = qc + i * QC_STRIDE;
    const double* r = E + 9;
    const double* U = E + 12;
    const double dinv = E[18];
    const double* Ia = E + 19;
    const double* a = m.axis[i];
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    double pAi[6];
    if (m.has_child[i]) { const double* sl = slot[m.slot_up[i]]; for (int k = 0; k < 6; ++k) pAi[k] = sl[k]; }
    else { for (int k = 0; k < 6; ++k) pAi[k] = cbp[i][6 + k]; }
    double sp = 0;
    for (int k = 0; k < 3; ++k) sp += a[k] * pAi[o + k];
    const double ui = tau[i] - sp;
    uu[i] = ui;
    const int par = m.parent[i];
    if (par >= 0) {
      double pa[6], Iac[6], fp[6];
      sym6_mv(Ia, cbp[i], Iac);
      for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
      xform_force_T(E, r, pa, fp);
      double* sl = slot[m.slot_up[par]];
      if (m.first_contrib[i]) { for (int k = 0; k < 6; ++k) sl[k] = cbp[par][6 + k] + fp[k]; }
      else { for (int k = 0; k < 6; ++k) sl[k] += fp[k]; }
    }
  }
  for (int i = 0; i < N; ++i) {
    const double* E = qc + i * QC_STRIDE;
    const double* r = E + 9;
    const double* U = E + 12;
    const double dinv = E[18];
    double ap[6];
    const int par = m.parent[i];
    if (par >= 0) xform_motion(E, r, slot[m.slot_down[par]], ap);
    else {
      const double a0[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
      xform_motion(E, r, a0, ap);
    }
    double s = 0;
    for (int k = 0; k < 6; ++k) { ap[k] += cbp[i][k]; s += U[k] * ap[k]; }
    const double qd = (uu[i] - s) * dinv;
    qdd[i] = qd;
    if (m.has_child[i]) {
      const double* a = m.axis[i];
      const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
      double* sl = slot[m.slot_down[i]];
      for (int k = 0; k < 6; ++k) sl[k] = ap[k];
      sl[o] += a[0] * qd; sl[o + 1] += a[1] * qd; sl[o + 2] += a[2] * qd;
    }
  }
}

// eval_to (problem.hpp:441-461) with the q-dependent part of the ABA taken from `qc` (computed at x's configuration)
template <int NJ>
__device__ void eval_f_cached(const DevModel& m, const double* __restrict__ qc, const double* x, const double* u, double* x_out) {
  const int nv = m.nv;
  double acc[NJ];
  aba_vu_cached<NJ>(m, qc, x + nv, u, acc);
  for (int i = 0; i < nv; ++i) {
    const double vo = m.dt * x[nv + i];
    x_out[i] = x[i] + vo;
    x_out[nv + i] = x[nv + i] + acc[i] * m.dt;
  }
}

}  // namespace rbd

// ---- latency-oriented variant: per-evaluation state in LDS instead of scratch ------------------------------------
// The closed-loop rollouts of the forward sweep are a few hundred sequential chains: too few lanes to hide the
// latency of scratch (private HBM-backed) arrays.  Here the per-joint state of one evaluation lives in LDS,
// interleaved over the TPB lanes of the workgroup ([slot][lane], conflict free); same arithmetic as aba_tree.
namespace rbd {

constexpr int ABA_LDS_SLOTS = 59;   // per joint: E 9 | r 3 | cb 6 | pA 6 | IA 21 | U 6 | Dinv 1 | uu 1 | vel/acc 6

template <int NJ, int TPB>
__device__ void aba_tree_lds(const DevModel& m, const double* q, const double* v, const double* tau, double* qdd,
                             double* st, int lane) {
  const int N = m.nv;
  auto S = [&](int joint, int slot) -> double& { return st[(joint * ABA_LDS_SLOTS + slot) * TPB + lane]; };
  constexpr int oE = 0, oR = 9, oC = 12, oP = 18, oI = 24, oU = 45, oD = 51, oT = 52, oV = 53;
  for (int i = 0; i < N; ++i) {
    double E[9], R[3], vel[6], vp[6], cb[6], pA[6], Iv[6], I6[21];
    joint_placement(m, i, q[i], E, R);
    const double* a = m.axis[i];
    double vJ[6] = {0, 0, 0, 0, 0, 0};
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    vJ[o] = a[0] * v[i]; vJ[o + 1] = a[1] * v[i]; vJ[o + 2] = a[2] * v[i];
    const int par = m.parent[i];
    if (par >= 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) vp[k] = S(par, oV + k);
      xform_motion(E, R, vp, vel);
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) vel[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
    crm(vel, vJ, cb);
#pragma unroll
    for (int k = 0; k < 21; ++k) I6[k] = m.I6[i][k];
    sym6_mv(I6, vel, Iv);
    crf(vel, Iv, pA);
#pragma unroll
    for (int k = 0; k < 9; ++k) S(i, oE + k) = E[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) S(i, oR + k) = R[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { S(i, oV + k) = vel[k]; S(i, oC + k) = cb[k]; S(i, oP + k) = pA[k]; }
#pragma unroll
    for (int k = 0; k < 21; ++k) S(i, oI + k) = I6[k];
  }
  for (int i = N - 1; i >= 0; --i) {
    const double* a = m.axis[i];
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    double IA[21], U[6], pAi[6], cb[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] = S(i, oI + k);
#pragma unroll
    for (int k = 0; k < 6; ++k) { pAi[k] = S(i, oP + k); cb[k] = S(i, oC + k); }
    double d = 0, sp = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) U[r] = IA[sidx(r, o)] * a[0] + IA[sidx(r, o + 1)] * a[1] + IA[sidx(r, o + 2)] * a[2];
    for (int k = 0; k < 3; ++k) { d += a[k] * U[o + k]; sp += a[k] * pAi[o + k]; }
    const double dinv = 1.0 / d;
    const double ui = tau[i] - sp;
#pragma unroll
    for (int k = 0; k < 6; ++k) S(i, oU + k) = U[k];
    S(i, oD) = dinv;
    S(i, oT) = ui;
    const int par = m.parent[i];
    if (par >= 0) {
      double E[9], R[3], Ia[21], pa[6], Iac[6], fp[6], IAp[21];
#pragma unroll
      for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) Ia[sidx(r, c)] = IA[sidx(r, c)] - U[r] * U[c] * dinv;
      sym6_mv(Ia, cb, Iac);
#pragma unroll
      for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
#pragma unroll
      for (int k = 0; k < 21; ++k) IAp[k] = S(par, oI + k);
      add_xtix(E, R, Ia, IAp);
#pragma unroll
      for (int k = 0; k < 21; ++k) S(par, oI + k) = IAp[k];
      xform_force_T(E, R, pa, fp);
#pragma unroll
      for (int k = 0; k < 6; ++k) S(par, oP + k) += fp[k];
    }
  }
  for (int i = 0; i < N; ++i) {
    double E[9], R[3], ap[6], accp[6], U[6], cb[6];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
#pragma unroll
    for (int k = 0; k < 6; ++k) { U[k] = S(i, oU + k); cb[k] = S(i, oC + k); }
    const int par = m.parent[i];
    if (par >= 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) accp[k] = S(par, oV + k);
      xform_motion(E, R, accp, ap);
    } else {
      const double a0[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
      xform_motion(E, R, a0, ap);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; s += U[k] * ap[k]; }
    const double qd = (S(i, oT) - s) * S(i, oD);
    qdd[i] = qd;
    const double* a = m.axis[i];
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    ap[o] += a[0] * qd; ap[o + 1] += a[1] * qd; ap[o + 2] += a[2] * qd;
#pragma unroll
    for (int k = 0; k < 6; ++k) S(i, oV + k) = ap[k];
  }
}

template <int NJ, int TPB>
__device__ void eval_f_lds(const DevModel& m, const double* x, const double* u, double* x_out, double* st, int lane) {
  const int nv = m.nv;
  if (m.kind == DDP_HIP_MODEL_PENDULUM) { eval_f<NJ>(m, x, u, x_out); return; }
  double acc[NJ];
  aba_tree_lds<NJ, TPB>(m, x, x + nv, u, acc, st, lane);
  for (int i = 0; i < nv; ++i) {
    const double vo = m.dt * x[nv + i];
    x_out[i] = x[i] + vo;
    x_out[nv + i] = x[nv + i] + acc[i] * m.dt;
  }
}

}  // namespace rbd

// ---- second level of the split: what depends on (q, v) but not on tau ---------------------------------------------
namespace rbd {

constexpr int VC_STRIDE = 18;   // per joint: cb[6] | pA0[6] (bias force before the children's contributions) | Ia cb [6]

// pass 1 of aba_vu_cached + the product Ia cb of its pass 2, stored for every evaluation that shares (q, v)
template <int NJ>
__device__ void aba_vpart_cached(const DevModel& m, const double* __restrict__ qc, const double* v, double* __restrict__ vc) {
  const int N = m.nv;
  double vel[NJ][6];
  for (int i = 0; i < N; ++i) {
    const double* E = qc + i * QC_STRIDE;
    const double* r = E + 9;
    const double* Ia = E + 19;
    const double* a = m.axis[i];
    double vJ[6] = {0, 0, 0, 0, 0, 0};
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    vJ[o] = a[0] * v[i]; vJ[o + 1] = a[1] * v[i]; vJ[o + 2] = a[2] * v[i];
    const int par = m.parent[i];
    if (par >= 0) xform_motion(E, r, vel[par], vel[i]);
    else { for (int k = 0; k < 6; ++k) vel[i][k] = 0.0; }
    for (int k = 0; k < 6; ++k) vel[i][k] += vJ[k];
    double cb[6], pA[6], Iv[6], Iac[6];
    crm(vel[i], vJ, cb);
    sym6_mv(m.I6[i], vel[i], Iv);
    crf(vel[i], Iv, pA);
    sym6_mv(Ia, cb, Iac);
    double* o_ = vc + i * VC_STRIDE;
    for (int k = 0; k < 6; ++k) { o_[k] = cb[k]; o_[6 + k] = pA[k]; o_[12 + k] = Iac[k]; }
  }
}

// passes 2 and 3 of aba_vu_cached for a new tau, with O(tree width) private state: the per-joint sums live in a few
// slots (DevModel::slot_up / slot_down) instead of per-joint arrays; the additions happen in exactly the order of
// aba_vu_cached (a joint's bias force starts from its cached value when its largest-index child contributes).
// tau(i) is supplied by a functor so that the caller does not need a private copy of the control vector.
constexpr int MAX_SLOTS = 8;
template <int NJ, typename TauFn>
__device__ void aba_u_cached(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ vc, TauFn tau,
                             double* qdd) {
  const int N = m.nv;
  double slot[MAX_SLOTS][6], uu[NJ];
  for (int i = N - 1; i >= 0; --i) {
    const double* E = qc + i * QC_STRIDE;
    const double* r = E + 9;
    const double* U = E + 12;
    const double dinv = E[18];
    const double* pA0 = vc + i * VC_STRIDE + 6;
    const double* Iac = vc + i * VC_STRIDE + 12;
    const double* a = m.axis[i];
    const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
    double pAi[6];
    if (m.has_child[i]) { const double* sl = slot[m.slot_up[i]]; for (int k = 0; k < 6; ++k) pAi[k] = sl[k]; }
    else { for (int k = 0; k < 6; ++k) pAi[k] = pA0[k]; }
    double sp = 0;
    for (int k = 0; k < 3; ++k) sp += a[k] * pAi[o + k];
    const double ui = tau(i) - sp;
    uu[i] = ui;
    const int par = m.parent[i];
    if (par >= 0) {
      double pa[6], fp[6];
      for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
      xform_force_T(E, r, pa, fp);
      double* sl = slot[m.slot_up[par]];
      if (m.first_contrib[i]) { const double* pp0 = vc + par * VC_STRIDE + 6; for (int k = 0; k < 6; ++k) sl[k] = pp0[k] + fp[k]; }
      else { for (int k = 0; k < 6; ++k) sl[k] += fp[k]; }
    }
  }
  for (int i = 0; i < N; ++i) {
    const double* E = qc + i * QC_STRIDE;
    const double* r = E + 9;
    const double* U = E + 12;
    const double dinv = E[18];
    const double* cb = vc + i * VC_STRIDE;
    double ap[6];
    const int par = m.parent[i];
    if (par >= 0) xform_motion(E, r, slot[m.slot_down[par]], ap);
    else {
      const double a0[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
      xform_motion(E, r, a0, ap);
    }
    double s = 0;
    for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; s += U[k] * ap[k]; }
    const double qd = (uu[i] - s) * dinv;
    qdd[i] = qd;
    if (m.has_child[i]) {
      const double* a = m.axis[i];
      const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
      double* sl = slot[m.slot_down[i]];
      for (int k = 0; k < 6; ++k) sl[k] = ap[k];
      sl[o] += a[0] * qd; sl[o + 1] += a[1] * qd; sl[o + 2] += a[2] * qd;
    }
  }
}

template <int NJ>
__device__ void eval_f_ucached(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ vc, const double* x,
                               const double* u, double* x_out) {
  const int nv = m.nv;
  double acc[NJ];
  aba_u_cached<NJ>(m, qc, vc, [&](int i) { return u[i]; }, acc);
  for (int i = 0; i < nv; ++i) {
    const double vo = m.dt * x[nv + i];
    x_out[i] = x[i] + vo;
    x_out[nv + i] = x[nv + i] + acc[i] * m.dt;
  }
}

}  // namespace rbd

// ---- wave-cooperative variant: the lanes of one evaluation walk the tree level by level -----------------------------
// NH lanes share one evaluation; in every round lane h takes the h-th joint of the current tree level (the branches of a
// humanoid advance side by side: 16 rounds per pass instead of 38 joints).  State as in aba_tree_lds ([slot][candidate]
// in LDS).  A child leaves its contribution to the parent (X^T Ia X, X^T pa) in its own slots; the parent adds its
// children's contributions in descending index order, i.e. in exactly the order of the sequential loop.
// Must be called by all lanes of the workgroup (it contains workgroup barriers); q, v, tau, qdd live in LDS.
namespace rbd {

// What the cooperative traversal reads of a model, as a plain struct a kernel can keep in LDS (fwd.hip: the per-level
// reads of axis / I6 / parent / level tables are then LDS reads instead of dependent global loads on the critical path).
template <int NJ>
struct CoopModel {
  double I6[NJ][21];
  double axis[NJ][3];
  double Rp[NJ][9];
  double pp[NJ][3];
  double gravity[3];
  double dt, c;
  int32_t parent[NJ], jtype[NJ];
  int32_t lvl_start[NJ + 1], lvl_joint[NJ];
  int32_t child_start[NJ + 1], child_list[NJ];
  int32_t n_levels, nv;
  int32_t nj, pad_;                 // joints (nv - 5 with a free-flyer root, else nv)
  unsigned long long role[DDP_MAXJ > 16 * 16 ? 1 : 16 * 16];   // coop_role of (level L, helper lane h) at role[L * NH + h] (NH <= 16, <= 16 levels)
};

// a lane's joint of one tree level in one word: joint (255: none) | parent + 1 | revolute | #children | children (descending index)
constexpr int ROLE_MAX_CHILDREN = 3;
__host__ __device__ __forceinline__ unsigned long long coop_role(int joint, int parent, bool rev, int nch, const int* ch) {
  unsigned long long r = (unsigned long long)(joint & 255) | ((unsigned long long)((parent + 1) & 255) << 8) | ((unsigned long long)(rev ? 1 : 0) << 16) |
                         ((unsigned long long)(nch & 3) << 17);
  for (int c = 0; c < nch && c < ROLE_MAX_CHILDREN; ++c) r |= (unsigned long long)(ch[c] & 255) << (19 + 8 * c);
  return r;
}
__device__ __forceinline__ int role_joint(unsigned long long r) { return (int)(r & 255); }
__device__ __forceinline__ int role_parent(unsigned long long r) { return (int)((r >> 8) & 255) - 1; }
__device__ __forceinline__ bool role_rev(unsigned long long r) { return ((r >> 16) & 1) != 0; }
__device__ __forceinline__ int role_nchildren(unsigned long long r) { return (int)((r >> 17) & 3); }
__device__ __forceinline__ int role_child(unsigned long long r, int c) { return (int)((r >> (19 + 8 * c)) & 255); }

// WAVE_SYNC: the workgroup is one wave -- LDS operations of a wave are processed in order, so the exchange points only have
// to stop the compiler from moving LDS accesses across them (no s_barrier, and above all no vmcnt(0): global loads issued
// ahead of the traversal stay in flight through it)
template <bool WAVE_SYNC>
__device__ __forceinline__ void coop_sync() {
  if constexpr (WAVE_SYNC) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// development (-DFWD_STAMPS, tools/fwd_stamps.py): phase clock of the forward kernel
struct FwdStamp {
  unsigned long long last, acc[12];
  __device__ __forceinline__ void mark(int i) { const unsigned long long now = wall_clock64(); acc[i] += now - last; last = now; }
};
#ifdef FWD_STAMPS
#define FSTAMP(fs, i) do { if (fs) (fs)->mark(i); } while (0)
#else
#define FSTAMP(fs, i) do { } while (0)
#endif

template <int NJ, int TPB, int NH, bool WAVE_SYNC = false, class M = DevModel>
__device__ void aba_tree_coop(const M& m, const double* q, const double* v, const double* tau, double* qdd,
                              double* st, int cand, int h, bool live, FwdStamp* fs = nullptr) {
  auto S = [&](int joint, int slot) -> double& { return st[(joint * ABA_LDS_SLOTS + slot) * TPB + cand]; };
  constexpr int oE = 0, oR = 9, oC = 12, oP = 18, oI = 24, oU = 45, oD = 51, oT = 52, oV = 53;
  const int NL = m.n_levels;
  // joint placements depend on q alone: all joints at once, off the level-by-level critical path
  if (live)
    for (int i = h; i < m.nv; i += NH) {
      double E[9], R[3];
      joint_placement(m, i, q[i], E, R);
#pragma unroll
      for (int k = 0; k < 9; ++k) S(i, oE + k) = E[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) S(i, oR + k) = R[k];
    }
  coop_sync<WAVE_SYNC>();
  FSTAMP(fs, 3);
  for (int L = 0; L < NL; ++L) {                 // pass 1, root -> leaves
    const int idx = m.lvl_start[L] + h;
    if (live && idx < m.lvl_start[L + 1]) {
      const int i = m.lvl_joint[idx];
      double E[9], R[3], vel[6], vp[6], cb[6], pA[6], Iv[6], I6[21];
#pragma unroll
      for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
      const double* a = m.axis[i];
      double vJ[6] = {0, 0, 0, 0, 0, 0};
      const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
      vJ[o] = a[0] * v[i]; vJ[o + 1] = a[1] * v[i]; vJ[o + 2] = a[2] * v[i];
      const int par = m.parent[i];
      if (par >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vp[k] = S(par, oV + k);
        xform_motion(E, R, vp, vel);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) vel[k] = 0.0;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
      crm(vel, vJ, cb);
#pragma unroll
      for (int k = 0; k < 21; ++k) I6[k] = m.I6[i][k];
      sym6_mv(I6, vel, Iv);
      crf(vel, Iv, pA);
#pragma unroll
      for (int k = 0; k < 6; ++k) { S(i, oV + k) = vel[k]; S(i, oC + k) = cb[k]; S(i, oP + k) = pA[k]; }
#pragma unroll
      for (int k = 0; k < 21; ++k) S(i, oI + k) = I6[k];
    }
    coop_sync<WAVE_SYNC>();
  }
  FSTAMP(fs, 4);
  for (int L = NL - 1; L >= 0; --L) {            // pass 2, leaves -> root
    const int idx = m.lvl_start[L] + h;
    if (live && idx < m.lvl_start[L + 1]) {
      const int i = m.lvl_joint[idx];
      const double* a = m.axis[i];
      const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
      double IA[21], U[6], pAi[6], cb[6];
#pragma unroll
      for (int k = 0; k < 21; ++k) IA[k] = S(i, oI + k);
#pragma unroll
      for (int k = 0; k < 6; ++k) { pAi[k] = S(i, oP + k); cb[k] = S(i, oC + k); }
      for (int ci = m.child_start[i]; ci < m.child_start[i + 1]; ++ci) {   // contributions, descending child index
        const int c = m.child_list[ci];
#pragma unroll
        for (int k = 0; k < 21; ++k) IA[k] += S(c, oI + k);
#pragma unroll
        for (int k = 0; k < 6; ++k) pAi[k] += S(c, oP + k);
      }
      double d = 0, sp = 0;
      // o is per lane: static indices under a select (dynamic ones would put IA in scratch: a store / load round trip per level)
      const bool rev = o == 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double i0 = rev ? IA[sidx(r, 0)] : IA[sidx(r, 3)];
        const double i1 = rev ? IA[sidx(r, 1)] : IA[sidx(r, 4)];
        const double i2 = rev ? IA[sidx(r, 2)] : IA[sidx(r, 5)];
        U[r] = i0 * a[0] + i1 * a[1] + i2 * a[2];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double Uk = rev ? U[k] : U[3 + k];
        const double pk = rev ? pAi[k] : pAi[3 + k];
        d += a[k] * Uk;
        sp += a[k] * pk;
      }
      const double dinv = 1.0 / d;
      const double ui = tau[i] - sp;
#pragma unroll
      for (int k = 0; k < 6; ++k) S(i, oU + k) = U[k];
      S(i, oD) = dinv;
      S(i, oT) = ui;
      if (m.parent[i] >= 0) {
        double E[9], R[3], Ia[21], pa[6], Iac[6], fp[6], Z[21];
#pragma unroll
        for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) Ia[sidx(r, c)] = IA[sidx(r, c)] - U[r] * U[c] * dinv;
        sym6_mv(Ia, cb, Iac);
#pragma unroll
        for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
#pragma unroll
        for (int k = 0; k < 21; ++k) Z[k] = 0.0;
        add_xtix(E, R, Ia, Z);                    // this joint's contribution to its parent, left in its own slots
#pragma unroll
        for (int k = 0; k < 21; ++k) S(i, oI + k) = Z[k];
        xform_force_T(E, R, pa, fp);
#pragma unroll
        for (int k = 0; k < 6; ++k) S(i, oP + k) = fp[k];
      }
    }
    coop_sync<WAVE_SYNC>();
  }
  FSTAMP(fs, 5);
  for (int L = 0; L < NL; ++L) {                 // pass 3, root -> leaves
    const int idx = m.lvl_start[L] + h;
    if (live && idx < m.lvl_start[L + 1]) {
      const int i = m.lvl_joint[idx];
      double E[9], R[3], ap[6], accp[6], U[6], cb[6];
#pragma unroll
      for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
#pragma unroll
      for (int k = 0; k < 6; ++k) { U[k] = S(i, oU + k); cb[k] = S(i, oC + k); }
      const int par = m.parent[i];
      if (par >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) accp[k] = S(par, oV + k);
        xform_motion(E, R, accp, ap);
      } else {
        const double a0[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
        xform_motion(E, R, a0, ap);
      }
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; s += U[k] * ap[k]; }
      const double qd = (S(i, oT) - s) * S(i, oD);
      qdd[i] = qd;
      const double* a = m.axis[i];
      const int o = m.jtype[i] == DDP_HIP_JOINT_REVOLUTE ? 0 : 3;
      ap[o] += a[0] * qd; ap[o + 1] += a[1] * qd; ap[o + 2] += a[2] * qd;
#pragma unroll
      for (int k = 0; k < 6; ++k) S(i, oV + k) = ap[k];
    }
    coop_sync<WAVE_SYNC>();
  }
  FSTAMP(fs, 6);
}

// ---- two-wave variant (fwd.hip: forward_kernel_lat2) ------------------------------------------------------------------
// A lone wave issues one FP64 instruction per 8 clocks (half of what its SIMD sustains), and the leaf -> root pass is 700 of
// them per tree level: that pass is 54 % of a rollout step (in-kernel stamps).  Its two halves are independent once U, 1/D and
// the articulated inertia Ia are known: the force recursion (pa, X^T pa) and the inertia contribution X^T Ia X.  Here a second
// wave of the workgroup forms X^T Ia X while the first does the force half; both recompute the short common prefix (children's
// sums, U, 1/D) so that neither waits for the other inside a level; one workgroup barrier per level.  The joint's own inertia
// is read from the model table (not from a per-candidate copy), so the slots the second wave writes its contribution to (oZ) are
// read by nobody before the level's barrier.  Every entry is formed by the same operations in the same order as in aba_tree_coop.
constexpr int ABA_LDS_SLOTS2 = ABA_LDS_SLOTS;   // same record: the joint's own inertia now comes from the model table, its 21 slots hold the contribution
__device__ __forceinline__ void wg_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FF: joint 0 is a free-flyer root (SE(3); q = [p, quaternion x y z w | joint angles], v = [linear, angular | joint rates] in the
// body frame): S = identity on the body twist, its 6 x 6 articulated inertia is solved in registers (sym6_solve) -- the
// operations of aba_tree's free-flyer branches, in their order; joint i >= 1 reads q[i + 6], v[i + 5]
template <int NJ, int TPB, int NH, class M, bool FF = false>
__device__ __forceinline__ void aba_tree_coop2w(const M& m, const double* q, const double* v, const double* tau, double* qdd,
                                double* st, int cand, int h, bool live, int wave, FwdStamp* fs = nullptr) {
  const int nj = FF ? m.nv - 5 : m.nv;
  auto QI = [](int i) { return FF ? i + 6 : i; };
  auto VI = [](int i) { return FF ? i + 5 : i; };
  // joints eight apart (the two arms and the head of a humanoid advance side by side) would share LDS banks: skew the records
  auto S = [&](int joint, int slot) -> double& { return st[((joint + (joint >> 3)) * ABA_LDS_SLOTS2 + slot) * TPB + cand]; };
  constexpr int oE = 0, oR = 9, oC = 12, oP = 18, oZ = 24, oU = 45, oD = 51, oT = 52, oV = 53;
  const int NL = m.n_levels;
  // m.role[L * NH + h]: everything lane h needs to know about its joint of level L in one LDS word (coop_role): the level loops
  // otherwise walk lvl_start -> lvl_joint -> parent / child_start -> child_list, four dependent LDS round trips per level and pass
  if (wave == 0) {
    if (live)
      for (int i = h; i < nj; i += NH) {
        double E[9], R[3];
        if (FF && i == 0) {                          // rbd::place: E = R(quaternion)^T, r = p
          double Rq[9];
          lie::quat_to_R(q + 3, Rq);
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int l = 0; l < 3; ++l) E[3 * k + l] = Rq[3 * l + k];
          R[0] = q[0]; R[1] = q[1]; R[2] = q[2];
        } else {
          joint_placement(m, i, q[QI(i)], E, R);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) S(i, oE + k) = E[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) S(i, oR + k) = R[k];
      }
    coop_sync<true>();
    FSTAMP(fs, 3);
    for (int L = 0; L < NL; ++L) {                 // pass 1, root -> leaves
      const unsigned long long rr = m.role[L * NH + h];
      if (live && role_joint(rr) != 255) {
        const int i = role_joint(rr);
        double E[9], R[3], vel[6], vp[6], cb[6], pA[6], Iv[6], I6[21];
#pragma unroll
        for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
        const double* a = m.axis[i];
        double vJ[6] = {0, 0, 0, 0, 0, 0};
        const int o = role_rev(rr) ? 0 : 3;
        if (FF && i == 0) { vJ[0] = v[3]; vJ[1] = v[4]; vJ[2] = v[5]; vJ[3] = v[0]; vJ[4] = v[1]; vJ[5] = v[2]; }   // S = identity on the body twist
        else { const double vi = v[VI(i)]; vJ[o] = a[0] * vi; vJ[o + 1] = a[1] * vi; vJ[o + 2] = a[2] * vi; }
        const int par = role_parent(rr);
        if (par >= 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) vp[k] = S(par, oV + k);
          xform_motion(E, R, vp, vel);
        } else {
#pragma unroll
          for (int k = 0; k < 6; ++k) vel[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
        crm(vel, vJ, cb);
#pragma unroll
        for (int k = 0; k < 21; ++k) I6[k] = m.I6[i][k];
        sym6_mv(I6, vel, Iv);
        crf(vel, Iv, pA);
#pragma unroll
        for (int k = 0; k < 6; ++k) { S(i, oV + k) = vel[k]; S(i, oC + k) = cb[k]; S(i, oP + k) = pA[k]; }
      }
      coop_sync<true>();
    }
    FSTAMP(fs, 4);
  }
  wg_sync_lds();                                   // placements (and the first wave's pass 1) are in LDS
  for (int L = NL - 1; L >= 0; --L) {              // pass 2, leaves -> root: wave 0 the forces, wave 1 the inertias
    const unsigned long long rr = m.role[L * NH + h];
    if (live && role_joint(rr) != 255) {
      const int i = role_joint(rr);
      const double* a = m.axis[i];
      const bool rev = role_rev(rr);
      const int nch = role_nchildren(rr);
      double IA[21], U[6];
#pragma unroll
      for (int k = 0; k < 21; ++k) IA[k] = m.I6[i][k];
#pragma unroll
      for (int ci = 0; ci < ROLE_MAX_CHILDREN; ++ci) {   // contributions, descending child index
        if (ci < nch) {
          const int c = role_child(rr, ci);
#pragma unroll
          for (int k = 0; k < 21; ++k) IA[k] += S(c, oZ + k);
        }
      }
      FSTAMP(fs, 9);
      if (FF && i == 0) {
        // the free-flyer root is resolved in the last pass: its articulated inertia (wave 1) and bias force (wave 0) stay in
        // its record (aba_tree: IAr, pAr); it has no parent to contribute to
        if (wave == 0) {
          double pAi[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) pAi[k] = S(i, oP + k);
#pragma unroll
          for (int ci = 0; ci < ROLE_MAX_CHILDREN; ++ci) {
            if (ci < nch) {
              const int c = role_child(rr, ci);
#pragma unroll
              for (int k = 0; k < 6; ++k) pAi[k] += S(c, oP + k);
            }
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) S(i, oP + k) = pAi[k];
        } else {
#pragma unroll
          for (int k = 0; k < 21; ++k) S(i, oZ + k) = IA[k];
        }
      } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double i0 = rev ? IA[sidx(r, 0)] : IA[sidx(r, 3)];
        const double i1 = rev ? IA[sidx(r, 1)] : IA[sidx(r, 4)];
        const double i2 = rev ? IA[sidx(r, 2)] : IA[sidx(r, 5)];
        U[r] = i0 * a[0] + i1 * a[1] + i2 * a[2];
      }
      double d = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) d += a[k] * (rev ? U[k] : U[3 + k]);
      const double dinv = 1.0 / d;
      FSTAMP(fs, 10);
      const bool has_parent = role_parent(rr) >= 0;
      if (wave == 0) {
        double pAi[6], cb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { pAi[k] = S(i, oP + k); cb[k] = S(i, oC + k); }
#pragma unroll
        for (int ci = 0; ci < ROLE_MAX_CHILDREN; ++ci) {
          if (ci < nch) {
            const int c = role_child(rr, ci);
#pragma unroll
            for (int k = 0; k < 6; ++k) pAi[k] += S(c, oP + k);
          }
        }
        double sp = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) sp += a[k] * (rev ? pAi[k] : pAi[3 + k]);
        const double ui = tau[VI(i)] - sp;
#pragma unroll
        for (int k = 0; k < 6; ++k) S(i, oU + k) = U[k];
        S(i, oD) = dinv;
        S(i, oT) = ui;
        if (has_parent) {
          double E[9], R[3], Ia[21], pa[6], Iac[6], fp[6];
#pragma unroll
          for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
          for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) Ia[sidx(r, c)] = IA[sidx(r, c)] - U[r] * U[c] * dinv;
          sym6_mv(Ia, cb, Iac);
#pragma unroll
          for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
          xform_force_T(E, R, pa, fp);
#pragma unroll
          for (int k = 0; k < 6; ++k) S(i, oP + k) = fp[k];
        }
      } else if (has_parent) {
        double E[9], R[3], Ia[21], Z[21];
#pragma unroll
        for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) Ia[sidx(r, c)] = IA[sidx(r, c)] - U[r] * U[c] * dinv;
#pragma unroll
        for (int k = 0; k < 21; ++k) Z[k] = 0.0;
        add_xtix(E, R, Ia, Z);
#pragma unroll
        for (int k = 0; k < 21; ++k) S(i, oZ + k) = Z[k];
      }
      }   // (not the free-flyer root)
    }
    FSTAMP(fs, 11);
    wg_sync_lds();
    FSTAMP(fs, 5);
  }
  if (wave != 0) return;
  FSTAMP(fs, 5);
  for (int L = 0; L < NL; ++L) {                   // pass 3, root -> leaves
    const unsigned long long rr = m.role[L * NH + h];
    if (live && role_joint(rr) != 255) {
      const int i = role_joint(rr);
      double E[9], R[3], ap[6], accp[6], U[6], cb[6];
#pragma unroll
      for (int k = 0; k < 9; ++k) E[k] = S(i, oE + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) R[k] = S(i, oR + k);
#pragma unroll
      for (int k = 0; k < 6; ++k) { U[k] = S(i, oU + k); cb[k] = S(i, oC + k); }
      const int par = role_parent(rr);
      if (par >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) accp[k] = S(par, oV + k);
        xform_motion(E, R, accp, ap);
      } else {
        const double a0[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
        xform_motion(E, R, a0, ap);
      }
      if (FF && i == 0) {
        // S = I: qdd_s = IA^-1 (tau_s - pA) - a' in the spatial ordering [angular; linear]; tau / qdd are ordered [linear; angular]
        double IAr[21], rhs[6], qs[6];
#pragma unroll
        for (int k = 0; k < 21; ++k) IAr[k] = S(i, oZ + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) { rhs[k] = tau[3 + k] - S(i, oP + k); rhs[3 + k] = tau[k] - S(i, oP + 3 + k); }
        sym6_solve(IAr, rhs, qs);
#pragma unroll
        for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; qs[k] -= ap[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { qdd[k] = qs[3 + k]; qdd[3 + k] = qs[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) S(i, oV + k) = ap[k] + qs[k];
      } else {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; s += U[k] * ap[k]; }
      const double qd = (S(i, oT) - s) * S(i, oD);
      qdd[VI(i)] = qd;
      const double* a = m.axis[i];
      const int o = role_rev(rr) ? 0 : 3;
      ap[o] += a[0] * qd; ap[o + 1] += a[1] * qd; ap[o + 2] += a[2] * qd;
#pragma unroll
      for (int k = 0; k < 6; ++k) S(i, oV + k) = ap[k];
      }   // (not the free-flyer root)
    }
    coop_sync<true>();
  }
  FSTAMP(fs, 6);
}

}  // namespace rbd
