// rbd_deriv.h -- analytic partial derivatives of the forward dynamics of a tree of 1-DoF joints: what the reference's
// first_order_deriv (problem.hpp:463-503) takes from model_t::d_dynamics_aba (pinocchio_model.ipp:359-400, i.e.
// Pinocchio's computeABADerivatives; Pinocchio is absent, the recursion is restated from Carpentier & Mansard,
// "Analytical derivatives of rigid body dynamics algorithms", RSS 2018):
//     d qdd/dq = -M^-1 d tau/dq,   d qdd/dv = -M^-1 d tau/dv,   d qdd/d tau = M^-1,     tau = RNEA(q, v, qdd)
// with the partials of the inverse dynamics formed in WORLD coordinates.  J_i: world-frame axis of joint i (a spatial
// motion vector [angular; linear]); ov, oa: world-frame body velocities / accelerations (gravity folded into a_0);
// I_k: world-frame body inertia; h = I ov; of = I oa + ov x* h; B_k x = I_k (x x ov_k) + x x* h_k + ov_k x* (I_k x);
// Ic, Bc, ofc: sums over the subtree.  Then
//     u_j = J_j x ov_j,   g_j = u_j x ov_j - J_j x oa_j
//     i in path(j):           d tau_i/dq_j = J_i . (J_j x* ofc_j - Bc_j u_j + Ic_j g_j)
//                             d tau_i/dv_j = J_i . (Bc_j J_j - 2 Ic_j u_j),      M_ij = J_i . Ic_j J_j
//     j proper ancestor of i: d tau_i/dq_j = -(Bc_i^T J_i) . u_j + (Ic_i J_i) . g_j
//                             d tau_i/dv_j =  (Bc_i^T J_i) . J_j - 2 (Ic_i J_i) . u_j
// (derivation: DESIGN.md section 4b).  This header holds the pieces shared by the one-lane-per-evaluation variant
// (small models: the constraint chain and mode 1 of the UR5-like drivers) and the wave-per-evaluation kernels
// (lin_analytic.hip, the Talos-like tree).
#pragma once

#include "rbd.h"

namespace rbdd {

using rbd::cross3;
using rbd::crf;
using rbd::crm;
using rbd::mm3;
using rbd::mv3;

// world placement of joint i from its parent's: oR (row-major, world = oR * joint coordinates), op
__device__ __forceinline__ void world_placement(const DevModel& m, int i, double q, const double* oRp, const double* opp,
                                                double* oR, double* op) {
  double E[9], r[3], Rc[9];
  rbd::joint_placement(m, i, q, E, r);
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = 0; l < 3; ++l) Rc[3 * k + l] = E[3 * l + k];
  if (oRp) {
    double t[3];
    mm3(oRp, Rc, oR);
    mv3(oRp, r, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) op[k] = opp[k] + t[k];
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) oR[k] = Rc[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) op[k] = r[k];
  }
}

// J_i: the joint axis as a world-frame spatial motion vector (the linear part is the velocity of the body point that
// coincides with the world origin)
__device__ __forceinline__ void world_axis(const DevModel& m, int i, const double* oR, const double* op, double* J) {
  double aw[3];
  mv3(oR, m.axis[i], aw);
  if (m.jtype[i] == DDP_HIP_JOINT_REVOLUTE) {
    double t[3];
    cross3(op, aw, t);
    J[0] = aw[0]; J[1] = aw[1]; J[2] = aw[2]; J[3] = t[0]; J[4] = t[1]; J[5] = t[2];
  } else {
    J[0] = 0.0; J[1] = 0.0; J[2] = 0.0; J[3] = aw[0]; J[4] = aw[1]; J[5] = aw[2];
  }
}

// y = A x for a row-major 6 x 6 (ld 6), y = A^T x
__device__ __forceinline__ void m6v(const double* A, const double* x, double* y) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += A[6 * r + c] * x[c];
    y[r] = s;
  }
}
__device__ __forceinline__ void m6tv(const double* A, const double* x, double* y) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double s = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += A[6 * r + c] * x[r];
    y[c] = s;
  }
}

// world-frame spatial inertia (about the world origin) of the body of joint i: full 6 x 6, row-major.
// I6b is the body-frame inertia packed as in DevModel::I6 (lower triangle): I_world = X^-T I_body X^-1 is formed from
// the mass, the world-frame centre of mass and the rotated rotational inertia about the centre of mass.
__device__ __forceinline__ void world_inertia(const double* I6b, const double* oR, const double* op, double* I6) {
  // body frame: I6b = [Io, m cx; m cx^T, m 1] with Io about the joint origin; mass = I6b(3,3); m c = (I6b(2,4)... )
  const double mass = I6b[rbd::sidx(3, 3)];
  // m cx = [[0,-mcz,mcy],[mcz,0,-mcx],[-mcy,mcx,0]] stored at rows 0..2, cols 3..5
  const double mc[3] = {I6b[rbd::sidx(2, 4)], I6b[rbd::sidx(0, 5)], I6b[rbd::sidx(1, 3)]};   // (m cx)(2,1), (0,2), (1,0)
  double Io[9];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = 0; l < 3; ++l) Io[3 * k + l] = I6b[rbd::sidx(k, l)];
  // rotate: Io_w = oR Io oR^T (about the joint origin, world axes); m c_w = oR (m c)
  double T1[9], Iw[9], mcw[3];
  mm3(oR, Io, T1);
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = 0; l < 3; ++l) Iw[3 * k + l] = T1[3 * k] * oR[3 * l] + T1[3 * k + 1] * oR[3 * l + 1] + T1[3 * k + 2] * oR[3 * l + 2];
  mv3(oR, mc, mcw);
  // shift the reference point from the joint origin to the world origin: with d = op (joint origin in world coordinates),
  // m c_O = m c_w + m d;  I_O = I_w + (m c_w)x dx^T ... written out: I_O = I_w - [mcw]x [d]x - [d]x [mcw]x - m [d]x [d]x
  const double d[3] = {op[0], op[1], op[2]};
  const double mco[3] = {mcw[0] + mass * d[0], mcw[1] + mass * d[1], mcw[2] + mass * d[2]};
  const double cxw[9] = {0, -mcw[2], mcw[1], mcw[2], 0, -mcw[0], -mcw[1], mcw[0], 0};
  const double dx[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
  double A1[9], A2[9], A3[9];
  mm3(cxw, dx, A1);
  mm3(dx, cxw, A2);
  mm3(dx, dx, A3);
#pragma unroll
  for (int k = 0; k < 36; ++k) I6[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = 0; l < 3; ++l) I6[6 * k + l] = Iw[3 * k + l] - A1[3 * k + l] - A2[3 * k + l] - mass * A3[3 * k + l];
  const double cx[9] = {0, -mco[2], mco[1], mco[2], 0, -mco[0], -mco[1], mco[0], 0};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = 0; l < 3; ++l) { I6[6 * k + l + 3] = cx[3 * k + l]; I6[6 * (k + 3) + l] = cx[3 * l + k]; }
  I6[21] = mass; I6[28] = mass; I6[35] = mass;
}

// B x = I (x x ov) + x x* h + ov x* (I x), column by column (row-major 6 x 6)
__device__ __forceinline__ void bias_matrix(const double* I6, const double* ov, const double* h, double* B) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double e[6] = {0, 0, 0, 0, 0, 0}, exv[6], t1[6], t2[6], Icol[6], t3[6];
    e[c] = 1.0;
    crm(e, ov, exv);
    m6v(I6, exv, t1);
    crf(e, h, t2);
#pragma unroll
    for (int k = 0; k < 6; ++k) Icol[k] = I6[6 * k + c];
    crf(ov, Icol, t3);
#pragma unroll
    for (int k = 0; k < 6; ++k) B[6 * k + c] = t1[k] + t2[k] + t3[k];
  }
}

// Cholesky M = L L^T in place on the lower triangle (column-major, ld), then Minv = M^-1 (full, column-major, ld):
// the columns of the identity through the two substitutions.  One lane does everything: small models only.
template <int NJ>
__device__ void chol_inverse_lane(int N, double* M, double* Minv) {
  for (int k = 0; k < N; ++k) {
    double x = M[k + k * NJ];
    for (int j = 0; j < k; ++j) x -= M[k + j * NJ] * M[k + j * NJ];
    x = sqrt(x);
    M[k + k * NJ] = x;
    for (int i = k + 1; i < N; ++i) {
      double s = M[i + k * NJ];
      for (int j = 0; j < k; ++j) s -= M[i + j * NJ] * M[k + j * NJ];
      M[i + k * NJ] = s / x;
    }
  }
  for (int c = 0; c < N; ++c) {
    double* b = Minv + c * NJ;
    for (int i = 0; i < N; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int l = 0; l < i; ++l) s -= M[i + l * NJ] * b[l];
      b[i] = s / M[i + i * NJ];
    }
    for (int i = N - 1; i >= 0; --i) {
      double s = b[i];
      for (int l = i + 1; l < N; ++l) s -= M[l + i * NJ] * b[l];
      b[i] = s / M[i + i * NJ];
    }
  }
}

// One lane: partials of qdd = ABA(q, v, tau).  dq, dv, dtau: nv x nv column-major with leading dimension nv.
// Private state O(NJ * 100) doubles: meant for NJ <= 6.
template <int NJ>
__device__ void aba_derivatives_lane(const DevModel& m, const double* q, const double* v, const double* tau, double* qdd,
                                     double* dq, double* dv, double* dtau) {
  const int N = m.nv;
  rbd::aba_tree<NJ>(m, q, v, tau, qdd);
  double oR[NJ][9], op[NJ][3], J[NJ][6], ov[NJ][6], oa[NJ][6], Ic[NJ][36], Bc[NJ][36], ofc[NJ][6];
  for (int i = 0; i < N; ++i) {
    const int par = m.parent[i];
    world_placement(m, i, q[i], par >= 0 ? oR[par] : nullptr, par >= 0 ? op[par] : nullptr, oR[i], op[i]);
    world_axis(m, i, oR[i], op[i], J[i]);
    double vJ[6], t6[6];
    for (int k = 0; k < 6; ++k) vJ[k] = J[i][k] * v[i];
    for (int k = 0; k < 6; ++k) ov[i][k] = (par >= 0 ? ov[par][k] : 0.0) + vJ[k];
    crm(ov[i], vJ, t6);
    for (int k = 0; k < 6; ++k) {
      const double ap = par >= 0 ? oa[par][k] : (k < 3 ? 0.0 : -m.gravity[k - 3]);
      oa[i][k] = ap + J[i][k] * qdd[i] + t6[k];
    }
    world_inertia(m.I6[i], oR[i], op[i], Ic[i]);
    double h[6], Ioa[6], vxh[6];
    m6v(Ic[i], ov[i], h);
    m6v(Ic[i], oa[i], Ioa);
    crf(ov[i], h, vxh);
    for (int k = 0; k < 6; ++k) ofc[i][k] = Ioa[k] + vxh[k];
    bias_matrix(Ic[i], ov[i], h, Bc[i]);
  }
  for (int i = N - 1; i >= 0; --i) {
    const int par = m.parent[i];
    if (par < 0) continue;
    for (int k = 0; k < 36; ++k) { Ic[par][k] += Ic[i][k]; Bc[par][k] += Bc[i][k]; }
    for (int k = 0; k < 6; ++k) ofc[par][k] += ofc[i][k];
  }
  double y[NJ][6], z[NJ][6], u[NJ][6], g[NJ][6], Fq[NJ][6], Fv[NJ][6];
  for (int i = 0; i < N; ++i) {
    double t1[6], t2[6], t3[6];
    m6v(Ic[i], J[i], y[i]);
    m6tv(Bc[i], J[i], z[i]);
    crm(J[i], ov[i], u[i]);
    crm(u[i], ov[i], t1);
    crm(J[i], oa[i], t2);
    for (int k = 0; k < 6; ++k) g[i][k] = t1[k] - t2[k];
    crf(J[i], ofc[i], t1);
    m6v(Bc[i], u[i], t2);
    m6v(Ic[i], g[i], t3);
    for (int k = 0; k < 6; ++k) Fq[i][k] = t1[k] - t2[k] + t3[k];
    m6v(Bc[i], J[i], t1);
    m6v(Ic[i], u[i], t2);
    for (int k = 0; k < 6; ++k) Fv[i][k] = t1[k] - 2.0 * t2[k];
  }
  double M[NJ * NJ], Tq[NJ * NJ], Tv[NJ * NJ], Minv[NJ * NJ];
  for (int k = 0; k < NJ * NJ; ++k) { M[k] = 0.0; Tq[k] = 0.0; Tv[k] = 0.0; }
  for (int j = 0; j < N; ++j) {
    for (int i = j; i >= 0; i = m.parent[i]) {
      double sq = 0, sv = 0, sm = 0;
      for (int k = 0; k < 6; ++k) { sq += J[i][k] * Fq[j][k]; sv += J[i][k] * Fv[j][k]; sm += J[i][k] * y[j][k]; }
      Tq[i + j * NJ] = sq; Tv[i + j * NJ] = sv;
      M[i + j * NJ] = sm; M[j + i * NJ] = sm;
    }
    for (int a = m.parent[j]; a >= 0; a = m.parent[a]) {
      double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
      for (int k = 0; k < 6; ++k) { s1 += z[j][k] * u[a][k]; s2 += y[j][k] * g[a][k]; s3 += z[j][k] * J[a][k]; s4 += y[j][k] * u[a][k]; }
      Tq[j + a * NJ] = -s1 + s2;
      Tv[j + a * NJ] = s3 - 2.0 * s4;
    }
  }
  chol_inverse_lane<NJ>(N, M, Minv);
  for (int c = 0; c < N; ++c)
    for (int r = 0; r < N; ++r) {
      double sq = 0, sv = 0;
      for (int l = 0; l < N; ++l) { sq += Minv[r + l * NJ] * Tq[l + c * NJ]; sv += Minv[r + l * NJ] * Tv[l + c * NJ]; }
      dq[r + c * N] = -sq;
      dv[r + c * N] = -sv;
      dtau[r + c * N] = Minv[r + c * NJ];
    }
}

// first_order_deriv (problem.hpp:463-503) on a vector space: fx = [I, dt I; dt da/dq, I + dt da/dv], fu = [0; dt da/dtau]
template <int NJ>
__device__ void first_order_analytic_lane(const DevModel& m, const double* x, const double* u, double* fx, double* fu, double* f) {
  const int nv = m.nv, n = 2 * nv;
  double qdd[NJ], dq[NJ * NJ], dv[NJ * NJ], dt_[NJ * NJ];
  aba_derivatives_lane<NJ>(m, x, x + nv, u, qdd, dq, dv, dt_);
  for (int i = 0; i < nv; ++i) {                       // eval_to (:441-461)
    const double vo = m.dt * x[nv + i];
    f[i] = x[i] + vo;
    f[nv + i] = x[nv + i] + qdd[i] * m.dt;
  }
  for (int k = 0; k < n * n; ++k) fx[k] = 0.0;
  for (int k = 0; k < n * nv; ++k) fu[k] = 0.0;
  for (int j = 0; j < nv; ++j) {
    fx[j + j * n] = 1.0;
    fx[j + (nv + j) * n] = 1.0 * m.dt;
    for (int i = 0; i < nv; ++i) {
      fx[(nv + i) + j * n] = dq[i + j * nv] * m.dt;
      fx[(nv + i) + (nv + j) * n] = dv[i + j * nv] * m.dt + (i == j ? 1.0 : 0.0);
      fu[(nv + i) + j * n] = dt_[i + j * nv] * m.dt;
    }
  }
}

}  // namespace rbdd
