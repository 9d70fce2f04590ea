/*
 * ddp_oracle.c -- CPU restatement of the reference DDP hot path.  TEST INFRASTRUCTURE ONLY
 * (see ddp_oracle.h for the rules and the parity status).  Plain C99, double, no dependencies.
 */
#include "ddp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers, column-major                                                          */
/* ------------------------------------------------------------------------------------------ */

static double* dalloc(int64_t n) {
  double* p = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
  return p;
}
static double* dzalloc(int64_t n) {
  double* p = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  return p;
}

/* C(r x c) += A(k x r)^T * B(k x c) */
static void gemm_tn_add(int64_t r, int64_t c, int64_t k, const double* A, int64_t lda, const double* B,
                        int64_t ldb, double* C, int64_t ldc) {
  for (int64_t j = 0; j < c; ++j)
    for (int64_t i = 0; i < r; ++i) {
      double s = 0.0;
      for (int64_t l = 0; l < k; ++l) s += A[l + i * lda] * B[l + j * ldb];
      C[i + j * ldc] += s;
    }
}
/* C(r x c) += A(r x k) * B(k x c) */
static void gemm_nn_add(int64_t r, int64_t c, int64_t k, const double* A, int64_t lda, const double* B,
                        int64_t ldb, double* C, int64_t ldc) {
  for (int64_t j = 0; j < c; ++j)
    for (int64_t i = 0; i < r; ++i) {
      double s = 0.0;
      for (int64_t l = 0; l < k; ++l) s += A[i + l * lda] * B[l + j * ldb];
      C[i + j * ldc] += s;
    }
}
/* y(r) += A(k x r)^T x(k) */
static void gemv_t_add(int64_t r, int64_t k, const double* A, int64_t lda, const double* x, double* y) {
  gemm_tn_add(r, 1, k, A, lda, x, k > 0 ? k : 1, y, r > 0 ? r : 1);
}
/* y(r) += A(r x k) x(k) */
static void gemv_n_add(int64_t r, int64_t k, const double* A, int64_t lda, const double* x, double* y) {
  gemm_nn_add(r, 1, k, A, lda, x, k > 0 ? k : 1, y, r > 0 ? r : 1);
}

/* tensor_view_t::noalias_contract_add_outdim, detail/tensor.hpp:179-198:
 * out(j,k) += sum_i v(i) T(i,j,k) with T(i,j,k) at i + j*O + k*O*L (tensor.hpp:146) */
static void contract_add_outdim(int64_t O, int64_t L, int64_t R, const double* Tn, const double* v, double* out) {
  for (int64_t jk = 0; jk < L * R; ++jk) {
    double s = 0.0;
    for (int64_t i = 0; i < O; ++i) s += v[i] * Tn[i + jk * O];
    out[jk] += s;
  }
}

/* Eigen::LLT<Lower> (unblocked left-looking form), in place on the lower triangle; returns -1 on
 * success or the index of the first non-positive pivot (ddp_bwd.ipp:104-105: info()==NumericalIssue) */
static int64_t llt_lower(int64_t m, double* A, int64_t lda) {
  for (int64_t k = 0; k < m; ++k) {
    double x = A[k + k * lda];
    for (int64_t j = 0; j < k; ++j) x -= A[k + j * lda] * A[k + j * lda];
    if (x <= 0.0) return k;
    x = sqrt(x);
    A[k + k * lda] = x;
    for (int64_t i = k + 1; i < m; ++i) {
      double s = A[i + k * lda];
      for (int64_t j = 0; j < k; ++j) s -= A[i + j * lda] * A[k + j * lda];
      A[i + k * lda] = s / x;
    }
  }
  return -1;
}
/* B(m x c) <- (L L^T)^{-1} B */
static void llt_solve(int64_t m, const double* Lm, int64_t lda, int64_t c, double* B, int64_t ldb) {
  for (int64_t j = 0; j < c; ++j) {
    double* b = B + j * ldb;
    for (int64_t i = 0; i < m; ++i) {
      double s = b[i];
      for (int64_t l = 0; l < i; ++l) s -= Lm[i + l * lda] * b[l];
      b[i] = s / Lm[i + i * lda];
    }
    for (int64_t i = m - 1; i >= 0; --i) {
      double s = b[i];
      for (int64_t l = i + 1; l < m; ++l) s -= Lm[l + i * lda] * b[l];
      b[i] = s / Lm[i + i * lda];
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* dims                                                                                       */
/* ------------------------------------------------------------------------------------------ */

static int m_nq(const orc_model* m);
int64_t orc_nx(const orc_problem* p) { return (int64_t)m_nq(&p->model) + p->model.nv; }
int64_t orc_ndx(const orc_problem* p) { return 2 * (int64_t)p->model.nv; }
int64_t orc_nu(const orc_problem* p) { return (int64_t)p->model.nv; }
int64_t orc_ne_total(const orc_problem* p) {
  int64_t s = 0;
  for (int64_t t = 0; t < p->T; ++t) s += p->ne[t];
  return s;
}
static int64_t ne_prefix(const int64_t* ne, int64_t t) {
  int64_t s = 0;
  for (int64_t i = 0; i < t; ++i) s += ne[i];
  return s;
}

/* ------------------------------------------------------------------------------------------ */
/* rigid body dynamics for trees of 1-DoF joints (restating what the reference delegates to   */
/* Pinocchio: pinocchio_model.ipp:353-355 aba).  Featherstone, RBDA, Table 7.1 / 5.1.         */
/* Spatial vectors are [angular; linear]; 6x6 matrices row-major.                             */
/* ------------------------------------------------------------------------------------------ */

static void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void mat3_mul(const double* A, const double* B, double* C) { /* row-major */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mat3_vec(const double* A, const double* x, double* y) {
  for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
static void skew(const double* a, double* S) {
  S[0] = 0; S[1] = -a[2]; S[2] = a[1];
  S[3] = a[2]; S[4] = 0; S[5] = -a[0];
  S[6] = -a[1]; S[7] = a[0]; S[8] = 0;
}
/* Rodrigues rotation about the unit axis a by angle q */
static void rot_axis(const double* a, double q, double* R) {
  double s = sin(q), c = cos(q), K[9], K2[9];
  skew(a, K);
  mat3_mul(K, K, K2);
  for (int i = 0; i < 9; ++i) R[i] = s * K[i] + (1.0 - c) * K2[i];
  R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}

/* ------------------------------------------------------------------------------------------ */
/* Lie-group configurations: a free-flyer root joint (SE(3); q = [p(3), quaternion x y z w],     */
/* v = [linear(3), angular(3)] in the body frame: Pinocchio's JointModelFreeFlyer), what           */
/* model_t::integrate / difference / d_integrate_dq,dv / d_difference_dq_start,finish              */
/* (pinocchio_model.ipp:222-321) delegate to pinocchio::integrate / difference / dIntegrate /      */
/* dDifference.  Pinocchio is absent: restated from the closed forms of the exponential and        */
/* logarithm of SE(3) and their Jacobians (Murray-Li-Sastry; Barfoot, State Estimation for         */
/* Robotics, eqs. 7.85-7.86 for the Q block).  Pinned by the reference's own property tests        */
/* (test/pinocchio.cpp:17-57,59-100: tests/test_lie.py) and by mpmath differences.                  */
/* Only joint 0 may be a free flyer (jtype ORC_JOINT_FREEFLYER, parent -1); then nq = nv + 1 and    */
/* the other joints j >= 1 use q[j + 6], v[j + 5].                                                  */
/* ------------------------------------------------------------------------------------------ */
static void inv6(const double* A, double* Ainv);
static int m_ff(const orc_model* m) { return m->kind == ORC_MODEL_TREE && m->jtype && m->jtype[0] == ORC_JOINT_FREEFLYER; }
static int m_nj(const orc_model* m) { return m_ff(m) ? m->nv - 5 : m->nv; }
static int m_nq(const orc_model* m) { return m_ff(m) ? m->nv + 1 : m->nv; }
static int m_qi(const orc_model* m, int j) { return m_ff(m) ? j + 6 : j; }   /* j >= 1 when ff */
static int m_vi(const orc_model* m, int j) { return m_ff(m) ? j + 5 : j; }
int32_t orc_model_nq(const orc_model* m) { return m_nq(m); }

static void quat_to_R(const double* qt, double* R) {      /* x y z w, unit; row-major */
  double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static void quat_mul(const double* a, const double* b, double* c) {
  double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
  c[0] = aw * bx + ax * bw + ay * bz - az * by;
  c[1] = aw * by - ax * bz + ay * bw + az * bx;
  c[2] = aw * bz + ax * by - ay * bx + az * bw;
  c[3] = aw * bw - ax * bx - ay * by - az * bz;
}
/* coefficients of the SO(3) / SE(3) series in theta = |w|:
 * a = sin t / t, b = (1 - cos t) / t^2, c = (t - sin t) / t^3, d = (1 - (t/2) cot(t/2)) / t^2.
 * The closed forms cancel near 0 (relative error ~ 6 eps / t^2 for c, 12 eps / t^2 for d: round 2 switched to them at
 * t = 1e-4, where that is 2e-8), so the Taylor series runs up to t^2 < 0.04 (six terms, truncation < 1e-16) and the closed
 * forms, written on the half angle, take over where they are good to 1e-13.  Pinned against mpmath in tests/test_lie.py. */
static void so3_coeffs(double t2, double* a, double* b, double* c, double* d) {
  if (t2 < 0.04) {
    *a = 1 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880 + t2 * (-1.0 / 39916800)))));
    *b = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 + t2 * (-1.0 / 479001600)))));
    *c = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 + t2 * (-1.0 / 6227020800.0)))));
    *d = 1.0 / 12 + t2 * (1.0 / 720 + t2 * (1.0 / 30240 + t2 * (1.0 / 1209600 + t2 * (1.0 / 47900160 + t2 * (691.0 / 1307674368000.0)))));
  } else {
    double t = sqrt(t2), st = sin(t), sh = sin(0.5 * t), ch = cos(0.5 * t);
    *a = st / t; *b = 2.0 * sh * sh / t2; *c = (t - st) / (t2 * t); *d = (1 - 0.5 * t * ch / sh) / t2;
  }
}
/* c4 = (1 - t^2/2 - cos t) / t^4, c6 = (t - sin t - t^3/6) / t^5 (Barfoot's Q block): cancellation ~ 24 eps / t^4 resp.
 * 120 eps / t^4, so the series (eight terms) runs up to t^2 < 1 */
static void so3_coeffs_q(double t2, double* c4, double* c6) {
  if (t2 < 1.0) {
    *c4 = -1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 + t2 * (-1.0 / 479001600 + t2 * (1.0 / 87178291200.0 +
          t2 * (-1.0 / 20922789888000.0 + t2 * (1.0 / 6402373705728000.0)))))));
    *c6 = -1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 + t2 * (-1.0 / 6227020800.0 + t2 * (1.0 / 1307674368000.0 +
          t2 * (-1.0 / 355687428096000.0 + t2 * (1.0 / 121645100408832000.0)))))));
  } else {
    double t = sqrt(t2);
    *c4 = (1 - t2 / 2 - cos(t)) / (t2 * t2); *c6 = (t - sin(t) - t2 * t / 6) / (t2 * t2 * t);
  }
}
/* test hook: out = {a, b, c, d, c4, c6} at t^2 */
void orc_so3_coeffs(double t2, double* out6) {
  so3_coeffs(t2, &out6[0], &out6[1], &out6[2], &out6[3]);
  so3_coeffs_q(t2, &out6[4], &out6[5]);
}
/* quaternion of exp3(w) */
static void quat_exp(const double* w, double* qt) {
  double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], k, cw;
  if (t2 < 1e-8) { k = 0.5 - t2 / 48; cw = 1 - t2 / 8 + t2 * t2 / 384; }
  else { double t = sqrt(t2); k = sin(t / 2) / t; cw = cos(t / 2); }
  qt[0] = k * w[0]; qt[1] = k * w[1]; qt[2] = k * w[2]; qt[3] = cw;
}
/* log3 of a unit quaternion (shortest rotation) */
static void quat_log(const double* qin, double* w) {
  double qt[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (qt[3] < 0) { qt[0] = -qt[0]; qt[1] = -qt[1]; qt[2] = -qt[2]; qt[3] = -qt[3]; }
  double n2 = qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2], k;
  if (n2 < 1e-16) k = 2.0 / qt[3] * (1 - n2 / (3 * qt[3] * qt[3]));
  else { double nn = sqrt(n2); k = 2 * atan2(nn, qt[3]) / nn; }
  w[0] = k * qt[0]; w[1] = k * qt[1]; w[2] = k * qt[2];
}
/* y = (I + alpha [w]x + beta [w]x^2) x */
static void so3_apply(const double* w, double alpha, double beta, const double* x, double* y) {
  double wx[3], wwx[3];
  cross3(w, x, wx);
  cross3(w, wx, wwx);
  for (int k = 0; k < 3; ++k) y[k] = x[k] + alpha * wx[k] + beta * wwx[k];
}
/* SE(3): q' = q (+) nu, nu = (v, w) body twist [linear; angular] (pinocchio SpecialEuclideanOperation<3>::integrate) */
static void se3_integrate(const double* q7, const double* nu, double* out7) {
  double R[9], a, b, c, d, pe[3], Rpe[3], qe[4], qn[4];
  const double* v = nu; const double* w = nu + 3;
  double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  so3_coeffs(t2, &a, &b, &c, &d);
  so3_apply(w, b, c, v, pe);                         /* V(w) v */
  quat_to_R(q7 + 3, R);
  mat3_vec(R, pe, Rpe);
  for (int k = 0; k < 3; ++k) out7[k] = q7[k] + Rpe[k];
  quat_exp(w, qe);
  quat_mul(q7 + 3, qe, qn);
  if (qn[0] * q7[3] + qn[1] * q7[4] + qn[2] * q7[5] + qn[3] * q7[6] < 0) for (int k = 0; k < 4; ++k) qn[k] = -qn[k];
  double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
  for (int k = 0; k < 4; ++k) out7[3 + k] = qn[k] / nn;
}
/* nu = q1 (-) q0 = log6(M0^-1 M1) */
static void se3_difference(const double* q0, const double* q1, double* nu) {
  double R0[9], dp[3], rp[3], q0c[4] = {-q0[3], -q0[4], -q0[5], q0[6]}, qr[4], w[3], a, b, c, d;
  quat_mul(q0c, q1 + 3, qr);
  quat_log(qr, w);
  quat_to_R(q0 + 3, R0);
  for (int k = 0; k < 3; ++k) dp[k] = q1[k] - q0[k];
  for (int k = 0; k < 3; ++k) rp[k] = R0[k] * dp[0] + R0[3 + k] * dp[1] + R0[6 + k] * dp[2];   /* R0^T dp */
  so3_coeffs(w[0] * w[0] + w[1] * w[1] + w[2] * w[2], &a, &b, &c, &d);
  so3_apply(w, -0.5, d, rp, nu);                     /* V(w)^-1 = I - 1/2 [w]x + d [w]x^2 */
  nu[3] = w[0]; nu[4] = w[1]; nu[5] = w[2];
}
/* Jacobian of log6 at M = exp6(nu) w.r.t. a right (body) perturbation of M: d(q1 (-) q0)/d q1 in the tangent at q1
 * (pinocchio dDifference ARG1 = Jlog6).  Row-major 6 x 6, rows / columns ordered [linear; angular]:
 *   Jlog6 = [ Jr^-1(w)   -Jr^-1(w) Q Jr^-1(w) ;  0   Jr^-1(w) ],  Jr^-1(w) = I + 1/2 [w]x + d [w]x^2,  Q = Q_r(v, w) */
static void se3_Jlog(const double* nu, double* J) {
  const double* v = nu; const double* w = nu + 3;
  double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], a, b, c, d;
  so3_coeffs(t2, &a, &b, &c, &d);
  double W[9], V[9], W2[9], Ji[9], Q[9], T1[9], T2[9], T3[9], T4[9];
  skew(w, W); skew(v, V);
  mat3_mul(W, W, W2);
  for (int k = 0; k < 9; ++k) Ji[k] = 0.5 * W[k] + d * W2[k];
  Ji[0] += 1; Ji[4] += 1; Ji[8] += 1;
  /* Barfoot's Q_l(rho, phi); the right-perturbation version is Q_l(-rho, -phi) */
  double c4, c6, c5;   /* (1 - t^2/2 - cos t)/t^4 ,  c5 = (c4 - 3 (t - sin t - t^3/6)/t^5)/2 */
  so3_coeffs_q(t2, &c4, &c6);
  c5 = 0.5 * (c4 - 3 * c6);
  double nW[9], nV[9], WV[9], VW[9], WVW[9], WWV[9], VWW[9], WVWW[9], WWVW[9];
  for (int k = 0; k < 9; ++k) { nW[k] = -W[k]; nV[k] = -V[k]; }
  mat3_mul(nW, nV, WV); mat3_mul(nV, nW, VW);
  mat3_mul(WV, nW, WVW);
  mat3_mul(nW, WV, WWV); mat3_mul(VW, nW, VWW);
  mat3_mul(WVW, nW, WVWW); mat3_mul(nW, WVW, WWVW);
  for (int k = 0; k < 9; ++k)
    Q[k] = 0.5 * nV[k] + c * (WV[k] + VW[k] + WVW[k]) - c4 * (WWV[k] + VWW[k] - 3 * WVW[k]) - c5 * (WVWW[k] + WWVW[k]);
  (void)T3; (void)T4; (void)a; (void)b;
  mat3_mul(Ji, Q, T1); mat3_mul(T1, Ji, T2);
  memset(J, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { J[6 * i + j] = Ji[3 * i + j]; J[6 * (i + 3) + j + 3] = Ji[3 * i + j]; J[6 * i + j + 3] = -T2[3 * i + j]; }
}

/* model_t::integrate, pinocchio_model.ipp:222-236 */
void orc_integrate(const orc_model* m, const double* q, const double* v, double* out_q) {
  if (!m_ff(m)) { for (int i = 0; i < m->nv; ++i) out_q[i] = q[i] + v[i]; return; }
  se3_integrate(q, v, out_q);
  for (int j = 1; j < m_nj(m); ++j) out_q[j + 6] = q[j + 6] + v[j + 5];
}
/* model_t::difference, pinocchio_model.ipp:271-286: out_v = q_finish (-) q_start */
void orc_difference(const orc_model* m, const double* q_start, const double* q_finish, double* out_v) {
  if (!m_ff(m)) { for (int i = 0; i < m->nv; ++i) out_v[i] = q_finish[i] - q_start[i]; return; }
  se3_difference(q_start, q_finish, out_v);
  for (int j = 1; j < m_nj(m); ++j) out_v[j + 5] = q_finish[j + 6] - q_start[j + 6];
}
/* model_t::d_difference_dq_finish, pinocchio_model.ipp:306-321 (nv x nv column-major) */
void orc_d_difference_dq_finish(const orc_model* m, const double* q_start, const double* q_finish, double* out) {
  int nv = m->nv;
  memset(out, 0, sizeof(double) * (size_t)nv * nv);
  for (int i = 0; i < nv; ++i) out[i + (int64_t)i * nv] = 1.0;
  if (!m_ff(m)) return;
  double nu[6], J[36];
  se3_difference(q_start, q_finish, nu);
  se3_Jlog(nu, J);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) out[i + (int64_t)j * nv] = J[6 * i + j];
}
/* model_t::d_difference_dq_start, pinocchio_model.ipp:288-304: -Jlog6(M) Ad(M^-1), M = M0^-1 M1 = exp6(nu);
 * equivalently -Jl^-1(nu) = -Jlog6(-nu)... formed here as -Jlog6 evaluated at -nu transposed structure: the left Jacobian */
void orc_d_difference_dq_start(const orc_model* m, const double* q_start, const double* q_finish, double* out) {
  int nv = m->nv;
  memset(out, 0, sizeof(double) * (size_t)nv * nv);
  for (int i = 0; i < nv; ++i) out[i + (int64_t)i * nv] = -1.0;
  if (!m_ff(m)) return;
  /* d log6(M0^-1 M1)/d(M0 right perturbation) = -Jl^-1(nu), and Jl^-1(nu) = Jr^-1(-nu) */
  double nu[6], J[36];
  se3_difference(q_start, q_finish, nu);
  for (int k = 0; k < 6; ++k) nu[k] = -nu[k];
  se3_Jlog(nu, J);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) out[i + (int64_t)j * nv] = -J[6 * i + j];
}
/* model_t::d_integrate_dq / d_integrate_dv, pinocchio_model.ipp:238-269: d(q (+) v)/dq = Ad(exp6(v))^-1 and
 * d(q (+) v)/dv = Jr(v) = Jlog6(v)^-1, in the tangent at q (+) v */
static void inv6(const double* A, double* Ainv) {   /* Gauss-Jordan with partial pivoting, row-major 6 x 6 */
  double M[6][12];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { M[i][j] = A[6 * i + j]; M[i][6 + j] = i == j; }
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (piv != c) for (int j = 0; j < 12; ++j) { double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
    double dinv = 1.0 / M[c][c];
    for (int j = 0; j < 12; ++j) M[c][j] *= dinv;
    for (int r = 0; r < 6; ++r) if (r != c) { double f = M[r][c]; for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j]; }
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ainv[6 * i + j] = M[i][6 + j];
}
void orc_d_integrate_dv(const orc_model* m, const double* q, const double* v, double* out) {
  int nv = m->nv;
  (void)q;
  memset(out, 0, sizeof(double) * (size_t)nv * nv);
  for (int i = 0; i < nv; ++i) out[i + (int64_t)i * nv] = 1.0;
  if (!m_ff(m)) return;
  double J[36], Ji[36];
  se3_Jlog(v, J);
  inv6(J, Ji);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) out[i + (int64_t)j * nv] = Ji[6 * i + j];
}
void orc_d_integrate_dq(const orc_model* m, const double* q, const double* v, double* out) {
  int nv = m->nv;
  (void)q;
  memset(out, 0, sizeof(double) * (size_t)nv * nv);
  for (int i = 0; i < nv; ++i) out[i + (int64_t)i * nv] = 1.0;
  if (!m_ff(m)) return;
  /* Ad(exp6(v)^-1) for twists ordered [linear; angular]: [R^T, -R^T [p]x; 0, R^T] with (R, p) = exp6(v) */
  double a, b, c, d, pe[3], qe[4], R[9], px[9], RtP[9], Rt[9];
  const double* w = v + 3;
  so3_coeffs(w[0] * w[0] + w[1] * w[1] + w[2] * w[2], &a, &b, &c, &d);
  so3_apply(w, b, c, v, pe);
  quat_exp(w, qe);
  quat_to_R(qe, R);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
  skew(pe, px);
  mat3_mul(Rt, px, RtP);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      out[i + (int64_t)j * nv] = Rt[3 * i + j];
      out[(i + 3) + (int64_t)(j + 3) * nv] = Rt[3 * i + j];
      out[i + (int64_t)(j + 3) * nv] = -RtP[3 * i + j];
      out[(i + 3) + (int64_t)j * nv] = 0.0;
    }
}

/* joint i at position q: E = rotation parent->child coordinates, r = child origin in parent frame */
static void joint_placement(const orc_model* m, int i, double q, double* E, double* r) {
  const double* Rp = m->Rp + 9 * i;
  const double* pp = m->pp + 3 * i;
  const double* a = m->axis + 3 * i;
  double RJ[9], Rc[9];
  if (m->jtype[i] == ORC_JOINT_REVOLUTE) {
    rot_axis(a, q, RJ);
    mat3_mul(Rp, RJ, Rc); /* parent coords = Rc * child coords */
    r[0] = pp[0]; r[1] = pp[1]; r[2] = pp[2];
  } else {
    double d[3] = {a[0] * q, a[1] * q, a[2] * q}, Rd[3];
    memcpy(Rc, Rp, sizeof(Rc));
    mat3_vec(Rp, d, Rd);
    r[0] = pp[0] + Rd[0]; r[1] = pp[1] + Rd[1]; r[2] = pp[2] + Rd[2];
  }
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < 3; ++l) E[3 * k + l] = Rc[3 * l + k];
}
/* placement of joint i from the whole configuration vector: a free-flyer root reads (p, quat), the others one scalar */
static void place(const orc_model* m, int i, const double* q, double* E, double* r) {
  if (i == 0 && m_ff(m)) {
    double R[9];
    quat_to_R(q + 3, R);                       /* world coords = R * body coords */
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) E[3 * k + l] = R[3 * l + k];
    r[0] = q[0]; r[1] = q[1]; r[2] = q[2];
    return;
  }
  joint_placement(m, i, q[m_qi(m, i)], E, r);
}
/* Pluecker motion transform X = [E 0; -E rx E] (RBDA eq. 2.24) */
static void plucker(const double* E, const double* r, double* X) {
  double rx[9], Erx[9];
  skew(r, rx);
  mat3_mul(E, rx, Erx);
  memset(X, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      X[6 * i + j] = E[3 * i + j];
      X[6 * (i + 3) + (j + 3)] = E[3 * i + j];
      X[6 * (i + 3) + j] = -Erx[3 * i + j];
    }
}
static void mat6_vec(const double* A, const double* x, double* y) {
  for (int i = 0; i < 6; ++i) {
    double s = 0;
    for (int j = 0; j < 6; ++j) s += A[6 * i + j] * x[j];
    y[i] = s;
  }
}
static void mat6_tvec(const double* A, const double* x, double* y) {
  for (int i = 0; i < 6; ++i) {
    double s = 0;
    for (int j = 0; j < 6; ++j) s += A[6 * j + i] * x[j];
    y[i] = s;
  }
}
/* spatial inertia of body i in its own frame (RBDA eq. 2.63) */
static void body_inertia(const orc_model* m, int i, double* I6) {
  double cx[9], cxT[9], cc[9];
  const double* c = m->com + 3 * i;
  double mass = m->mass_j[i];
  skew(c, cx);
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < 3; ++l) cxT[3 * k + l] = cx[3 * l + k];
  mat3_mul(cx, cxT, cc);
  memset(I6, 0, 36 * sizeof(double));
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < 3; ++l) {
      I6[6 * k + l] = m->Ic[9 * i + 3 * k + l] + mass * cc[3 * k + l];
      I6[6 * k + (l + 3)] = mass * cx[3 * k + l];
      I6[6 * (k + 3) + l] = mass * cxT[3 * k + l];
    }
  I6[6 * 3 + 3] = mass; I6[6 * 4 + 4] = mass; I6[6 * 5 + 5] = mass;
}
static void motion_subspace(const orc_model* m, int i, double* S) {
  const double* a = m->axis + 3 * i;
  memset(S, 0, 6 * sizeof(double));
  if (m->jtype[i] == ORC_JOINT_REVOLUTE) { S[0] = a[0]; S[1] = a[1]; S[2] = a[2]; }
  else { S[3] = a[0]; S[4] = a[1]; S[5] = a[2]; }
}
/* v x m (motion cross motion), v x* f (motion cross force) */
static void crm(const double* v, const double* mm, double* out) {
  double a[3], b[3], c[3];
  cross3(v, mm, a);
  cross3(v + 3, mm, b);
  cross3(v, mm + 3, c);
  out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
  out[3] = b[0] + c[0]; out[4] = b[1] + c[1]; out[5] = b[2] + c[2];
}
static void crf(const double* v, const double* f, double* out) {
  double a[3], b[3], c[3];
  cross3(v, f, a);
  cross3(v + 3, f + 3, b);
  cross3(v, f + 3, c);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2];
  out[3] = c[0]; out[4] = c[1]; out[5] = c[2];
}

static void pendulum_aba(const orc_model* m, const double* q, const double* tau, double* qdd) {
  /* pendulum_model.hpp:105-114; g = 9.81 (:26) */
  qdd[0] = -9.81 / m->length * sin(q[0]) + tau[0] / m->mass;
}

void orc_aba(const orc_model* m, const double* q, const double* v, const double* tau, double* qdd) {
  if (m->kind == ORC_MODEL_PENDULUM) { pendulum_aba(m, q, tau, qdd); return; }
  int N = m_nj(m), ff = m_ff(m);
  double* X = dalloc(36 * N);  double* vel = dalloc(6 * N); double* cb = dalloc(6 * N);
  double* IA = dalloc(36 * N); double* pA = dalloc(6 * N);  double* U = dalloc(6 * N);
  double* D = dalloc(N);       double* uu = dalloc(N);      double* acc = dalloc(6 * N);
  double S[6], E[9], r[3], tmp6[6];
  for (int i = 0; i < N; ++i) {
    place(m, i, q, E, r);
    plucker(E, r, X + 36 * i);
    double vJ[6];
    if (i == 0 && ff) {                        /* S = identity on the body twist; v = [linear; angular] */
      for (int k = 0; k < 3; ++k) { vJ[k] = v[3 + k]; vJ[3 + k] = v[k]; }
    } else {
      motion_subspace(m, i, S);
      for (int k = 0; k < 6; ++k) vJ[k] = S[k] * v[m_vi(m, i)];
    }
    if (m->parent[i] >= 0) mat6_vec(X + 36 * i, vel + 6 * m->parent[i], vel + 6 * i);
    else memset(vel + 6 * i, 0, 6 * sizeof(double));
    for (int k = 0; k < 6; ++k) vel[6 * i + k] += vJ[k];
    crm(vel + 6 * i, vJ, cb + 6 * i);
    body_inertia(m, i, IA + 36 * i);
    mat6_vec(IA + 36 * i, vel + 6 * i, tmp6);
    crf(vel + 6 * i, tmp6, pA + 6 * i);
  }
  for (int i = N - 1; i >= 0; --i) {
    if (i == 0 && ff) break;                   /* the 6-DoF root is resolved in the last pass */
    motion_subspace(m, i, S);
    mat6_vec(IA + 36 * i, S, U + 6 * i);
    double d = 0, sp = 0;
    for (int k = 0; k < 6; ++k) { d += S[k] * U[6 * i + k]; sp += S[k] * pA[6 * i + k]; }
    D[i] = d;
    uu[i] = tau[m_vi(m, i)] - sp;
    int par = m->parent[i];
    if (par >= 0) {
      double Ia[36], pa[6], Iac[6], XtIa[36], XtIaX[36];
      for (int k = 0; k < 6; ++k)
        for (int l = 0; l < 6; ++l) Ia[6 * k + l] = IA[36 * i + 6 * k + l] - U[6 * i + k] * U[6 * i + l] / d;
      mat6_vec(Ia, cb + 6 * i, Iac);
      for (int k = 0; k < 6; ++k) pa[k] = pA[6 * i + k] + Iac[k] + U[6 * i + k] * uu[i] / d;
      const double* Xi = X + 36 * i;
      for (int k = 0; k < 6; ++k)
        for (int l = 0; l < 6; ++l) {
          double s = 0;
          for (int j = 0; j < 6; ++j) s += Xi[6 * j + k] * Ia[6 * j + l];
          XtIa[6 * k + l] = s;
        }
      for (int k = 0; k < 6; ++k)
        for (int l = 0; l < 6; ++l) {
          double s = 0;
          for (int j = 0; j < 6; ++j) s += XtIa[6 * k + j] * Xi[6 * j + l];
          XtIaX[6 * k + l] = s;
        }
      for (int k = 0; k < 36; ++k) IA[36 * par + k] += XtIaX[k];
      mat6_tvec(Xi, pa, tmp6);
      for (int k = 0; k < 6; ++k) pA[6 * par + k] += tmp6[k];
    }
  }
  for (int i = 0; i < N; ++i) {
    double ap[6];
    int par = m->parent[i];
    if (par >= 0) mat6_vec(X + 36 * i, acc + 6 * par, ap);
    else {
      double a0[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
      mat6_vec(X + 36 * i, a0, ap);
    }
    for (int k = 0; k < 6; ++k) ap[k] += cb[6 * i + k];
    if (i == 0 && ff) {
      /* S = I: qdd_s = IA^-1 (tau_s - pA) - a', spatial ordering [angular; linear]; tau, qdd ordered [linear; angular] */
      double rhs[6], Ainv[36], qs[6];
      for (int k = 0; k < 3; ++k) { rhs[k] = tau[3 + k] - pA[k]; rhs[3 + k] = tau[k] - pA[3 + k]; }
      inv6(IA, Ainv);
      mat6_vec(Ainv, rhs, qs);
      for (int k = 0; k < 6; ++k) { qs[k] -= ap[k]; acc[k] = ap[k] + qs[k]; }
      for (int k = 0; k < 3; ++k) { qdd[k] = qs[3 + k]; qdd[3 + k] = qs[k]; }
      continue;
    }
    double s = 0;
    for (int k = 0; k < 6; ++k) s += U[6 * i + k] * ap[k];
    qdd[m_vi(m, i)] = (uu[i] - s) / D[i];
    motion_subspace(m, i, S);
    for (int k = 0; k < 6; ++k) acc[6 * i + k] = ap[k] + S[k] * qdd[m_vi(m, i)];
  }
  free(X); free(vel); free(cb); free(IA); free(pA); free(U); free(D); free(uu); free(acc);
}

/* recursive Newton-Euler (RBDA Table 5.1): used only to cross-check orc_aba */
void orc_rnea(const orc_model* m, const double* q, const double* v, const double* a, double* tau) {
  if (m->kind == ORC_MODEL_PENDULUM) {
    tau[0] = m->mass * (a[0] + 9.81 / m->length * sin(q[0]));
    return;
  }
  int N = m_nj(m), ff = m_ff(m);
  double* X = dalloc(36 * N); double* vel = dalloc(6 * N); double* acc = dalloc(6 * N); double* f = dalloc(6 * N);
  double S[6], E[9], r[3], I6[36], t1[6], t2[6];
  for (int i = 0; i < N; ++i) {
    place(m, i, q, E, r);
    plucker(E, r, X + 36 * i);
    double vJ[6], aJ[6];
    if (i == 0 && ff) {
      for (int k = 0; k < 3; ++k) { vJ[k] = v[3 + k]; vJ[3 + k] = v[k]; aJ[k] = a[3 + k]; aJ[3 + k] = a[k]; }
    } else {
      motion_subspace(m, i, S);
      for (int k = 0; k < 6; ++k) { vJ[k] = S[k] * v[m_vi(m, i)]; aJ[k] = S[k] * a[m_vi(m, i)]; }
    }
    int par = m->parent[i];
    if (par >= 0) {
      mat6_vec(X + 36 * i, vel + 6 * par, vel + 6 * i);
      mat6_vec(X + 36 * i, acc + 6 * par, acc + 6 * i);
    } else {
      double a0[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
      memset(vel + 6 * i, 0, 6 * sizeof(double));
      mat6_vec(X + 36 * i, a0, acc + 6 * i);
    }
    for (int k = 0; k < 6; ++k) vel[6 * i + k] += vJ[k];
    crm(vel + 6 * i, vJ, t1);
    for (int k = 0; k < 6; ++k) acc[6 * i + k] += aJ[k] + t1[k];
    body_inertia(m, i, I6);
    mat6_vec(I6, acc + 6 * i, t1);
    mat6_vec(I6, vel + 6 * i, t2);
    crf(vel + 6 * i, t2, f + 6 * i);
    for (int k = 0; k < 6; ++k) f[6 * i + k] += t1[k];
  }
  for (int i = N - 1; i >= 0; --i) {
    if (i == 0 && ff) {
      for (int k = 0; k < 3; ++k) { tau[k] = f[3 + k]; tau[3 + k] = f[k]; }
    } else {
      motion_subspace(m, i, S);
      double s = 0;
      for (int k = 0; k < 6; ++k) s += S[k] * f[6 * i + k];
      tau[m_vi(m, i)] = s;
    }
    int par = m->parent[i];
    if (par >= 0) {
      mat6_tvec(X + 36 * i, f + 6 * i, t1);
      for (int k = 0; k < 6; ++k) f[6 * par + k] += t1[k];
    }
  }
  free(X); free(vel); free(acc); free(f);
}

void orc_crba(const orc_model* m, const double* q, double* M) {
  int N = m->nv;
  if (m->kind == ORC_MODEL_PENDULUM) { M[0] = m->mass; return; }
  orc_model mz = *m;
  mz.gravity[0] = mz.gravity[1] = mz.gravity[2] = 0.0;
  double* z = dzalloc(N); double* e = dzalloc(N);
  for (int j = 0; j < N; ++j) {
    e[j] = 1.0;
    orc_rnea(&mz, q, z, e, M + (int64_t)j * N);
    e[j] = 0.0;
  }
  free(z); free(e);
}

/* world placement of every joint frame: oR (row-major, world coords = oR * joint coords), op */
static void forward_kinematics(const orc_model* m, const double* q, double* oR, double* op) {
  for (int i = 0; i < m_nj(m); ++i) {
    double E[9], r[3], Rc[9];
    place(m, i, q, E, r);
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) Rc[3 * k + l] = E[3 * l + k];
    int par = m->parent[i];
    if (par >= 0) {
      double t[3];
      mat3_mul(oR + 9 * par, Rc, oR + 9 * i);
      mat3_vec(oR + 9 * par, r, t);
      for (int k = 0; k < 3; ++k) op[3 * i + k] = op[3 * par + k] + t[k];
    } else {
      memcpy(oR + 9 * i, Rc, sizeof(Rc));
      memcpy(op + 3 * i, r, 3 * sizeof(double));
    }
  }
}

/* model_t::frame_coordinates, pinocchio_model.ipp:418-430 (translation of oMf) */
void orc_frame_position(const orc_model* m, int32_t joint, const double* off, const double* q, double* p3) {
  int N = m_nj(m);
  double* oR = dalloc(9 * N); double* op = dalloc(3 * N);
  double t[3];
  forward_kinematics(m, q, oR, op);
  mat3_vec(oR + 9 * joint, off, t);
  for (int k = 0; k < 3; ++k) p3[k] = op[3 * joint + k] + t[k];
  free(oR); free(op);
}

/* model_t::d_frame_coordinates, pinocchio_model.ipp:433-462: top three rows (Pinocchio orders a
 * Motion as [linear; angular]) of getFrameJacobian(..., WORLD): the velocity of the body point that
 * coincides with the WORLD ORIGIN, not of the frame origin (reference quirk kept as is). */
void orc_frame_jacobian(const orc_model* m, int32_t joint, const double* off, const double* q,
                        int world_aligned, double* J) {
  int N = m_nj(m);
  double* oR = dalloc(9 * N); double* op = dalloc(3 * N);
  double p[3], t[3];
  forward_kinematics(m, q, oR, op);
  mat3_vec(oR + 9 * joint, off, t);
  for (int k = 0; k < 3; ++k) p[k] = op[3 * joint + k] + t[k];
  memset(J, 0, sizeof(double) * 3 * (size_t)m->nv);
  for (int j = joint; j >= 0; j = m->parent[j]) {
    double lever[3];
    for (int k = 0; k < 3; ++k) lever[k] = (world_aligned ? p[k] : 0.0) - op[3 * j + k];
    if (j == 0 && m_ff(m)) {
      /* free flyer: columns of oMi.act(S), S = identity on [linear; angular] body twists */
      for (int c = 0; c < 3; ++c) {
        double e[3] = {0, 0, 0}, aw[3], col[3];
        e[c] = 1.0;
        mat3_vec(oR, e, aw);
        for (int k = 0; k < 3; ++k) J[k + 3 * c] = aw[k];                 /* linear direction c */
        cross3(aw, lever, col);
        for (int k = 0; k < 3; ++k) J[k + 3 * (3 + c)] = col[k];           /* angular direction c */
      }
      continue;
    }
    double aw[3], col[3];
    mat3_vec(oR + 9 * j, m->axis + 3 * j, aw);
    if (m->jtype[j] == ORC_JOINT_REVOLUTE) cross3(aw, lever, col);
    else { col[0] = aw[0]; col[1] = aw[1]; col[2] = aw[2]; }
    for (int k = 0; k < 3; ++k) J[k + 3 * m_vi(m, j)] = col[k];
  }
  free(oR); free(op);
}

/* ------------------------------------------------------------------------------------------ */
/* analytic derivatives of the forward dynamics (what the reference delegates to Pinocchio's      */
/* computeABADerivatives, pinocchio_model.ipp:390-399; Pinocchio is absent, version unpinned).    */
/* Restated from the published recursion (Carpentier & Mansard, "Analytical derivatives of rigid   */
/* body dynamics algorithms", RSS 2018): d qdd/dq = -M^-1 d tau/dq, d qdd/dv = -M^-1 d tau/dv,     */
/* d qdd/d tau = M^-1, with the partials of the inverse dynamics tau = RNEA(q, v, qdd) formed in   */
/* WORLD coordinates.  With J_i the world-frame axis of joint i, ov / oa the world-frame body      */
/* velocities / accelerations (gravity folded into a_0), I_k the world-frame body inertias,        */
/* h = I ov, of = I oa + ov x* h, B_k x = I_k (x x ov_k) + x x* h_k + ov_k x* (I_k x), and composite  */
/* (subtree) sums Ic, Bc, ofc:                                                                       */
/*   u_j = J_j x ov_j,   g_j = u_j x ov_j - J_j x oa_j                                              */
/*   i in path(j):          d tau_i/dq_j = J_i . (J_j x* ofc_j - Bc_j u_j + Ic_j g_j)               */
/*                          d tau_i/dv_j = J_i . (Bc_j J_j - 2 Ic_j u_j),   M_ij = J_i . Ic_j J_j   */
/*   j proper ancestor of i: d tau_i/dq_j = -(Bc_i^T J_i) . u_j + (Ic_i J_i) . g_j                  */
/*                          d tau_i/dv_j =  (Bc_i^T J_i) . J_j - 2 (Ic_i J_i) . u_j                 */
/* Pinned by central differences of orc_rnea / orc_aba and by an mpmath restatement at 50 digits    */
/* (tests/test_oracle_pinning.py); the reference holds no numbers for it: parity unpinned.          */
/* ------------------------------------------------------------------------------------------ */
static void forward_kinematics(const orc_model* m, const double* q, double* oR, double* op);

void orc_rnea_derivatives(const orc_model* m, const double* q, const double* v, const double* a,
                          double* dtau_dq, double* dtau_dv, double* M) {
  int N = m->nv;
  if (m->kind == ORC_MODEL_PENDULUM) {       /* tau = m (a + g/l sin q): pendulum_model.hpp:105-130 */
    dtau_dq[0] = m->mass * 9.81 / m->length * cos(q[0]);
    dtau_dv[0] = 0.0;
    M[0] = m->mass;
    return;
  }
  double* oR = dalloc(9 * N); double* op = dalloc(3 * N);
  double* J = dalloc(6 * N); double* ov = dalloc(6 * N); double* oa = dalloc(6 * N);
  double* Ic = dalloc(36 * N); double* Bc = dalloc(36 * N); double* ofc = dalloc(6 * N);
  double* y = dalloc(6 * N); double* z = dalloc(6 * N); double* u = dalloc(6 * N); double* g = dalloc(6 * N);
  double* Fq = dalloc(6 * N); double* Fv = dalloc(6 * N);
  forward_kinematics(m, q, oR, op);
  for (int i = 0; i < N; ++i) {
    double aw[3], t[3], t6[6];
    mat3_vec(oR + 9 * i, m->axis + 3 * i, aw);
    double* Ji = J + 6 * i;
    if (m->jtype[i] == ORC_JOINT_REVOLUTE) {
      cross3(op + 3 * i, aw, t);               /* velocity of the body point at the world origin */
      Ji[0] = aw[0]; Ji[1] = aw[1]; Ji[2] = aw[2]; Ji[3] = t[0]; Ji[4] = t[1]; Ji[5] = t[2];
    } else {
      Ji[0] = Ji[1] = Ji[2] = 0.0; Ji[3] = aw[0]; Ji[4] = aw[1]; Ji[5] = aw[2];
    }
    int par = m->parent[i];
    double a0[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
    const double* vp = par >= 0 ? ov + 6 * par : NULL;
    const double* ap = par >= 0 ? oa + 6 * par : a0;
    double vJ[6];
    for (int k = 0; k < 6; ++k) vJ[k] = Ji[k] * v[i];
    for (int k = 0; k < 6; ++k) ov[6 * i + k] = (vp ? vp[k] : 0.0) + vJ[k];
    crm(ov + 6 * i, vJ, t6);                   /* ov_i x J_i qd_i (= ov_parent x J_i qd_i) */
    for (int k = 0; k < 6; ++k) oa[6 * i + k] = ap[k] + Ji[k] * a[i] + t6[k];
    /* world-frame inertia of body i about the world origin */
    double cw[3], Icw[9], Rt[9], T1[9], cx[9], cxT[9], cc[9];
    mat3_vec(oR + 9 * i, m->com + 3 * i, cw);
    for (int k = 0; k < 3; ++k) cw[k] += op[3 * i + k];
    for (int k = 0; k < 3; ++k) for (int l = 0; l < 3; ++l) Rt[3 * k + l] = oR[9 * i + 3 * l + k];
    mat3_mul(oR + 9 * i, m->Ic + 9 * i, T1);
    mat3_mul(T1, Rt, Icw);
    skew(cw, cx);
    for (int k = 0; k < 3; ++k) for (int l = 0; l < 3; ++l) cxT[3 * k + l] = cx[3 * l + k];
    mat3_mul(cx, cxT, cc);
    double* I6 = Ic + 36 * i;
    double mass = m->mass_j[i];
    memset(I6, 0, 36 * sizeof(double));
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) {
        I6[6 * k + l] = Icw[3 * k + l] + mass * cc[3 * k + l];
        I6[6 * k + l + 3] = mass * cx[3 * k + l];
        I6[6 * (k + 3) + l] = mass * cxT[3 * k + l];
      }
    I6[21] = I6[28] = I6[35] = mass;
    double h[6], Ioa[6], vxh[6];
    mat6_vec(I6, ov + 6 * i, h);
    mat6_vec(I6, oa + 6 * i, Ioa);
    crf(ov + 6 * i, h, vxh);
    for (int k = 0; k < 6; ++k) ofc[6 * i + k] = Ioa[k] + vxh[k];
    /* B x = I (x x ov) + x x* h + ov x* (I x), column by column */
    double* B = Bc + 36 * i;
    for (int c = 0; c < 6; ++c) {
      double e[6] = {0, 0, 0, 0, 0, 0}, exv[6], t1[6], t2[6], Icol[6], t3[6];
      e[c] = 1.0;
      crm(e, ov + 6 * i, exv);
      mat6_vec(I6, exv, t1);
      crf(e, h, t2);
      for (int k = 0; k < 6; ++k) Icol[k] = I6[6 * k + c];
      crf(ov + 6 * i, Icol, t3);
      for (int k = 0; k < 6; ++k) B[6 * k + c] = t1[k] + t2[k] + t3[k];
    }
  }
  for (int i = N - 1; i >= 0; --i) {           /* composite (subtree) sums */
    int par = m->parent[i];
    if (par < 0) continue;
    for (int k = 0; k < 36; ++k) { Ic[36 * par + k] += Ic[36 * i + k]; Bc[36 * par + k] += Bc[36 * i + k]; }
    for (int k = 0; k < 6; ++k) ofc[6 * par + k] += ofc[6 * i + k];
  }
  for (int i = 0; i < N; ++i) {
    const double* Ji = J + 6 * i;
    double t1[6], t2[6], t3[6];
    mat6_vec(Ic + 36 * i, Ji, y + 6 * i);
    mat6_tvec(Bc + 36 * i, Ji, z + 6 * i);
    crm(Ji, ov + 6 * i, u + 6 * i);
    crm(u + 6 * i, ov + 6 * i, t1);
    crm(Ji, oa + 6 * i, t2);
    for (int k = 0; k < 6; ++k) g[6 * i + k] = t1[k] - t2[k];
    crf(Ji, ofc + 6 * i, t1);
    mat6_vec(Bc + 36 * i, u + 6 * i, t2);
    mat6_vec(Ic + 36 * i, g + 6 * i, t3);
    for (int k = 0; k < 6; ++k) Fq[6 * i + k] = t1[k] - t2[k] + t3[k];
    mat6_vec(Bc + 36 * i, Ji, t1);
    mat6_vec(Ic + 36 * i, u + 6 * i, t2);
    for (int k = 0; k < 6; ++k) Fv[6 * i + k] = t1[k] - 2.0 * t2[k];
  }
  memset(dtau_dq, 0, sizeof(double) * (size_t)N * N);
  memset(dtau_dv, 0, sizeof(double) * (size_t)N * N);
  memset(M, 0, sizeof(double) * (size_t)N * N);
  for (int j = 0; j < N; ++j) {
    for (int i = j; i >= 0; i = m->parent[i]) {            /* i in path(j) */
      double sq = 0, sv = 0, sm = 0;
      for (int k = 0; k < 6; ++k) { sq += J[6 * i + k] * Fq[6 * j + k]; sv += J[6 * i + k] * Fv[6 * j + k]; sm += J[6 * i + k] * y[6 * j + k]; }
      dtau_dq[i + (int64_t)j * N] = sq;
      dtau_dv[i + (int64_t)j * N] = sv;
      M[i + (int64_t)j * N] = sm;
      M[j + (int64_t)i * N] = sm;
    }
    for (int a_ = m->parent[j]; a_ >= 0; a_ = m->parent[a_]) {   /* a_ proper ancestor of j: row j, column a_ */
      double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
      for (int k = 0; k < 6; ++k) {
        s1 += z[6 * j + k] * u[6 * a_ + k]; s2 += y[6 * j + k] * g[6 * a_ + k];
        s3 += z[6 * j + k] * J[6 * a_ + k]; s4 += y[6 * j + k] * u[6 * a_ + k];
      }
      dtau_dq[j + (int64_t)a_ * N] = -s1 + s2;
      dtau_dv[j + (int64_t)a_ * N] = s3 - 2.0 * s4;
    }
  }
  free(oR); free(op); free(J); free(ov); free(oa); free(Ic); free(Bc); free(ofc);
  free(y); free(z); free(u); free(g); free(Fq); free(Fv);
}

/* model_t::d_dynamics_aba, pinocchio_model.ipp:359-400: partials of qdd = ABA(q, v, tau), nv x nv column-major each */
void orc_aba_derivatives(const orc_model* m, const double* q, const double* v, const double* tau,
                         double* dq, double* dv, double* dtau) {
  int N = m->nv;
  double* a = dalloc(N);
  double* tq = dalloc((int64_t)N * N); double* tv = dalloc((int64_t)N * N); double* M = dalloc((int64_t)N * N);
  orc_aba(m, q, v, tau, a);
  orc_rnea_derivatives(m, q, v, a, tq, tv, M);
  /* M = L L^T; columns of M^-1, -M^-1 dtau/dq, -M^-1 dtau/dv by substitution */
  int64_t bad = llt_lower(N, M, N);
  (void)bad;                                   /* a joint-space inertia matrix is positive definite */
  memset(dtau, 0, sizeof(double) * (size_t)N * N);
  for (int i = 0; i < N; ++i) dtau[i + (int64_t)i * N] = 1.0;
  llt_solve(N, M, N, N, dtau, N);
  for (int64_t k = 0; k < (int64_t)N * N; ++k) { dq[k] = -tq[k]; dv[k] = -tv[k]; }
  llt_solve(N, M, N, N, dq, N);
  llt_solve(N, M, N, N, dv, N);
  free(a); free(tq); free(tv); free(M);
}

/* ------------------------------------------------------------------------------------------ */
/* dynamics_t : problem.hpp:343-525                                                            */
/* ------------------------------------------------------------------------------------------ */

/* dynamics_t::integrate_x (problem.hpp:395-401): x (+) dx = (q (+) dx_q, v + dx_v);  difference_out for states
 * (problem.hpp:403-412): (q1 (-) q0, v1 - v0) */
static void integrate_x(const orc_problem* p, const double* x, const double* dx, double* out) {
  int nv = p->model.nv, nq = m_nq(&p->model);
  orc_integrate(&p->model, x, dx, out);
  for (int i = 0; i < nv; ++i) out[nq + i] = x[nq + i] + dx[nv + i];
}
static void difference_x(const orc_problem* p, const double* x0, const double* x1, double* out) {
  int nv = p->model.nv, nq = m_nq(&p->model);
  orc_difference(&p->model, x0, x1, out);
  for (int i = 0; i < nv; ++i) out[nv + i] = x1[nq + i] - x0[nq + i];
}

/* dynamics_t::eval_to, problem.hpp:441-461 (semi-implicit Euler) */
void orc_eval_f(const orc_problem* p, const double* x, const double* u, double* x_out) {
  int nv = p->model.nv, nq = m_nq(&p->model);
  const double* q = x; const double* v = x + nq;
  double* q_out = x_out; double* v_out = x_out + nq;
  double* acc = dalloc(nv);
  for (int i = 0; i < nv; ++i) v_out[i] = p->dt * v[i];
  orc_integrate(&p->model, q, v_out, q_out);                  /* model.integrate (problem.hpp:452) */
  orc_aba(&p->model, q, v, u, acc);
  for (int i = 0; i < nv; ++i) v_out[i] = v[i] + acc[i] * p->dt;
  free(acc);
}

/* dynamics_t::first_order_deriv, problem.hpp:463-503.
 *  - first_order_fd = 0: analytic, as the reference -- the pendulum's closed form (pendulum_model.hpp:116-130) or, for
 *    tree models, orc_aba_derivatives above (the reference calls Pinocchio's computeABADerivatives);
 *  - first_order_fd = 1: the build's north star asks for forward finite differences: column j is
 *    difference_out(f(x,u), f(x (+) eps e_j, u)) / eps, eps = sqrt(DBL_EPSILON), perturbing with
 *    integrate_x / integrate_u exactly like problem.hpp:105-126 does for the second order. */
void orc_first_order_f(const orc_problem* p, const double* x, const double* u, double* fx, double* fu, double* f) {
  int nv = p->model.nv;
  int64_t n = 2 * nv, m = nv;
  orc_eval_f(p, x, u, f);
  if (!p->first_order_fd && p->model.kind == ORC_MODEL_PENDULUM) {
    /* closed-form partials (pendulum_model.hpp:116-130) */
    double aq = -9.81 / p->model.length * cos(x[0]);
    double av = 0.0, at = 1.0 / p->model.mass;
    fx[0 + 0 * 2] = 1.0;            /* d_integrate_dq */
    fx[0 + 1 * 2] = 1.0 * p->dt;    /* d_integrate_dv * dt */
    fx[1 + 0 * 2] = aq * p->dt;
    fx[1 + 1 * 2] = av * p->dt + 1.0;
    fu[0] = 0.0;
    fu[1] = at * p->dt;
    return;
  }
  if (!p->first_order_fd && !m_ff(&p->model)) {
    /* problem.hpp:463-503 with d_dynamics_aba (:495): fx = [dInt_dq, dt dInt_dv; dt da/dq, I + dt da/dv], fu = [0; dt da/dtau];
     * on a vector space d_integrate_dq = d_integrate_dv = I (pendulum_model.hpp:64-84 does the same) */
    double* dq = dalloc((int64_t)nv * nv); double* dv = dalloc((int64_t)nv * nv); double* dt_ = dalloc((int64_t)nv * nv);
    orc_aba_derivatives(&p->model, x, x + nv, u, dq, dv, dt_);
    memset(fx, 0, sizeof(double) * (size_t)(n * n));
    memset(fu, 0, sizeof(double) * (size_t)(n * m));
    for (int j = 0; j < nv; ++j) {
      fx[j + (int64_t)j * n] = 1.0;                           /* top left */
      fx[j + (int64_t)(nv + j) * n] = 1.0 * p->dt;            /* top right: d_integrate_dv * dt (:489-490) */
      for (int i = 0; i < nv; ++i) {
        fx[(nv + i) + (int64_t)j * n] = dq[i + (int64_t)j * nv] * p->dt;                                  /* :499 */
        fx[(nv + i) + (int64_t)(nv + j) * n] = dv[i + (int64_t)j * nv] * p->dt + (i == j ? 1.0 : 0.0);    /* :500-501 */
        fu[(nv + i) + (int64_t)j * n] = dt_[i + (int64_t)j * nv] * p->dt;                                 /* :502 */
      }
    }
    free(dq); free(dv); free(dt_);
    return;
  }
  double eps = sqrt(DBL_EPSILON);
  int64_t nx = orc_nx(p);
  double* xp = dalloc(nx); double* up = dalloc(m); double* fp = dalloc(nx); double* dx = dzalloc(n); double* df = dalloc(n);
  for (int64_t j = 0; j < n + m; ++j) {
    memcpy(xp, x, sizeof(double) * (size_t)nx);
    memcpy(up, u, sizeof(double) * (size_t)m);
    if (j < n) { dx[j] = eps; integrate_x(p, x, dx, xp); dx[j] = 0.0; } else up[j - n] = u[j - n] + eps;
    orc_eval_f(p, xp, up, fp);
    difference_x(p, f, fp, df);                               /* difference_out */
    double* col = (j < n) ? fx + j * n : fu + (j - n) * n;
    for (int64_t k = 0; k < n; ++k) col[k] = df[k] / eps;
  }
  free(xp); free(up); free(fp); free(dx); free(df);
}

/* ------------------------------------------------------------------------------------------ */
/* constraints: problem.hpp:527-870                                                            */
/* ------------------------------------------------------------------------------------------ */

static const double* eq_target_at(const orc_problem* p, int64_t t) { return p->eq_target + ne_prefix(p->ne, t); }

/* base constraint value at solver time t (unshifted index t + advance):
 * config_constraint_t::eval_to problem.hpp:792-806 (plain subtraction q - target, :785-790);
 * spatial_constraint_t::eval_to problem.hpp:679-689 */
static void eq_base_eval(const orc_problem* p, int64_t t, const double* x, double* out) {
  int64_t e = p->ne[t];
  if (e == 0) return;
  const double* tg = eq_target_at(p, t);
  if (p->eq_kind == ORC_EQ_CONFIG) {
    for (int64_t i = 0; i < e; ++i) out[i] = x[i] - tg[i];
  } else {
    double pos[3];
    orc_frame_position(&p->model, p->frame_joint, p->frame_off, x, pos);
    for (int64_t i = 0; i < e; ++i) out[i] = pos[i] - tg[i];
  }
}
/* base constraint jacobian wrt x (e x n); wrt u it is zero (problem.hpp:719-721, :842-844) */
static void eq_base_first_order(const orc_problem* p, int64_t t, const double* x, double* out_x, double* out) {
  int64_t e = p->ne[t];
  int nv = p->model.nv;
  int64_t n = 2 * nv;
  if (e == 0) return;
  eq_base_eval(p, t, x, out);
  memset(out_x, 0, sizeof(double) * (size_t)(e * n));
  if (p->eq_kind == ORC_EQ_CONFIG) {
    /* d_difference_dq_finish is the identity on a vector space (pendulum_model.hpp:97-103) */
    for (int64_t i = 0; i < e; ++i) out_x[i + i * e] = 1.0;
  } else {
    double* J = dalloc(3 * nv);
    orc_frame_jacobian(&p->model, p->frame_joint, p->frame_off, x, 0, J);
    for (int j = 0; j < nv; ++j)
      for (int64_t i = 0; i < e; ++i) out_x[i + j * e] = J[i + 3 * j];   /* columns in tangent (v) order */
    free(J);
  }
}

/* constraint_advance_time_t::eval_to, problem.hpp:563-567, applied `level` times; the SAME u is
 * used for every look-ahead dynamics step (Appendix C of SURVEY.md) */
static void eq_eval_level(const orc_problem* p, int level, int64_t t, const double* x, const double* u, double* out) {
  if (p->ne[t] == 0) return;
  if (level == 0) { eq_base_eval(p, t, x, out); return; }
  int64_t nx = orc_nx(p);
  double* xn = dalloc(nx);
  orc_eval_f(p, x, u, xn);
  eq_eval_level(p, level - 1, t, xn, u, out);
  free(xn);
}
void orc_eval_eq(const orc_problem* p, int64_t t, const double* x, const double* u, double* out) {
  if (p->eq_kind == ORC_EQ_NONE) return;
  eq_eval_level(p, p->eq_advance, t, x, u, out);
}

/* constraint_advance_time_t::first_order_deriv, problem.hpp:569-605 (chain rule :603-604) */
static void eq_first_order_level(const orc_problem* p, int level, int64_t t, const double* x, const double* u,
                                 double* out_x, double* out_u, double* out) {
  int64_t e = p->ne[t], n = orc_ndx(p), m = orc_nu(p), nx = orc_nx(p);
  if (e == 0) return;
  if (level == 0) {
    eq_base_first_order(p, t, x, out_x, out);
    memset(out_u, 0, sizeof(double) * (size_t)(e * m));
    return;
  }
  double* xn = dalloc(nx); double* fxn = dalloc(n * n); double* fun = dalloc(n * m);
  double* enx = dalloc(e * n); double* enu = dalloc(e * m);
  orc_first_order_f(p, x, u, fxn, fun, xn);
  eq_first_order_level(p, level - 1, t, xn, u, enx, enu, out);
  memset(out_x, 0, sizeof(double) * (size_t)(e * n));
  memset(out_u, 0, sizeof(double) * (size_t)(e * m));
  gemm_nn_add(e, n, n, enx, e, fxn, n, out_x, e);
  gemm_nn_add(e, m, n, enx, e, fun, n, out_u, e);
  free(xn); free(fxn); free(fun); free(enx); free(enu);
}

/* ------------------------------------------------------------------------------------------ */
/* finite_diff_hessian_compute: problem.hpp:50-341                                             */
/* ------------------------------------------------------------------------------------------ */

typedef struct fd_fn {
  const orc_problem* p;
  int is_eq;   /* 0: dynamics_t, 1: constraint chain */
  int64_t t;
  int64_t o;   /* output dimension */
} fd_fn;

static void fn_eval(const fd_fn* f, const double* x, const double* u, double* out) {
  if (f->is_eq) orc_eval_eq(f->p, f->t, x, u, out); else orc_eval_f(f->p, x, u, out);
}
static void fn_first_order(const fd_fn* f, const double* x, const double* u, double* ox, double* ou, double* out) {
  if (f->is_eq) eq_first_order_level(f->p, f->p->eq_advance, f->t, x, u, ox, ou, out);
  else orc_first_order_f(f->p, x, u, ox, ou, out);
}

#define TIDX(i, j, k, O, L) ((i) + (j) * (O) + (k) * (O) * (L))

/* second_order_deriv_1, problem.hpp:67-150 */
static void fd_second_order_1(const fd_fn* f, const double* x, const double* u, double* oxx, double* oux,
                              double* ouu, double* ox, double* ou, double* out) {
  int64_t o = f->o, n = orc_ndx(f->p), m = orc_nu(f->p);
  fn_first_order(f, x, u, ox, ou, out);
  if (o == 0) return;
  int64_t nx_ = orc_nx(f->p);
  double* fx_ = dalloc(o * n); double* fu_ = dalloc(o * m); double* out_ = dalloc(o > nx_ ? o : nx_);
  double* x_ = dalloc(nx_); double* u_ = dalloc(m); double* dx = dzalloc(n); double* du = dzalloc(m);
  double eps = sqrt(DBL_EPSILON);
  for (int64_t i = 0; i < n + m; ++i) {
    int at_x = i < n;
    int64_t idx = at_x ? i : i - n;
    double* in_var = at_x ? &dx[idx] : &du[idx];
    *in_var = eps;
    integrate_x(f->p, x, dx, x_);                          /* integrate_x */
    for (int64_t k = 0; k < m; ++k) u_[k] = u[k] + du[k];  /* integrate_u, problem.hpp:403 */
    fn_first_order(f, x_, u_, fx_, fu_, out_);
    if (at_x) {
      for (int64_t k = 0; k < o; ++k) {
        for (int64_t j = 0; j < n; ++j) oxx[TIDX(k, j, idx, o, n)] = (fx_[k + j * o] - ox[k + j * o]) / eps;
        for (int64_t j = 0; j < m; ++j) oux[TIDX(k, j, idx, o, m)] = (fu_[k + j * o] - ou[k + j * o]) / eps;
      }
    } else {
      for (int64_t k = 0; k < o; ++k)
        for (int64_t j = 0; j < m; ++j) ouu[TIDX(k, j, idx, o, m)] = (fu_[k + j * o] - ou[k + j * o]) / eps;
    }
    *in_var = 0;
  }
  free(fx_); free(fu_); free(out_); free(x_); free(u_); free(dx); free(du);
}

/* second_order_deriv_2, problem.hpp:152-298 */
static void fd_second_order_2(const fd_fn* f, const double* x, const double* u, double* oxx, double* oux,
                              double* ouu, double* ox, double* ou, double* out) {
  int64_t o = f->o, n = orc_ndx(f->p), m = orc_nu(f->p);
  fn_first_order(f, x, u, ox, ou, out);
  if (o == 0) return;
  const double* f0 = out;
  int64_t nx_ = orc_nx(f->p);
  int lie = !f->is_eq;                                   /* the dynamics' output is a state: difference_out is (-) on the group */
  double* f1 = dzalloc(o > nx_ ? o : nx_); double* df = dzalloc(o);
  double* x1 = dalloc(nx_); double* u1 = dalloc(m); double* dx = dzalloc(n); double* du = dzalloc(m);
  double eps = sqrt(sqrt(DBL_EPSILON));
  double eps2 = eps * eps;
  /* diagonal, :192-222 */
  for (int64_t i = 0; i < n + m; ++i) {
    int at_x = i < n;
    int64_t idx = at_x ? i : i - n;
    double* in_var = at_x ? &dx[idx] : &du[idx];
    const double* f_col = at_x ? ox + idx * o : ou + idx * o;
    double* tensor = at_x ? oxx : ouu;
    int64_t L = at_x ? n : m;
    *in_var = eps;
    integrate_x(f->p, x, dx, x1);
    for (int64_t k = 0; k < m; ++k) u1[k] = u[k] + du[k];
    fn_eval(f, x1, u1, f1);
    if (lie) difference_x(f->p, f0, f1, df);                 /* difference_out (problem.hpp:206) */
    else for (int64_t k = 0; k < o; ++k) df[k] = f1[k] - f0[k];
    for (int64_t k = 0; k < o; ++k) df[k] -= eps * f_col[k];
    for (int64_t k = 0; k < o; ++k) df[k] *= 2;
    for (int64_t k = 0; k < o; ++k) tensor[TIDX(k, idx, idx, o, L)] = df[k] / eps2;
    *in_var = 0;
  }
  /* off-diagonal, :226-296 */
  for (int64_t i = 0; i < n + m; ++i) {
    int at_x_1 = i < n;
    int64_t idx_1 = at_x_1 ? i : i - n;
    double* in_var_1 = at_x_1 ? &dx[idx_1] : &du[idx_1];
    const double* f_col_1 = at_x_1 ? ox + idx_1 * o : ou + idx_1 * o;
    const double* tensor_1 = at_x_1 ? oxx : ouu;
    int64_t L1 = at_x_1 ? n : m;
    *in_var_1 = eps;
    for (int64_t j = i + 1; j < n + m; ++j) {
      int at_x_2 = j < n;
      int64_t idx_2 = at_x_2 ? j : j - n;
      double* in_var_2 = at_x_2 ? &dx[idx_2] : &du[idx_2];
      const double* f_col_2 = at_x_2 ? ox + idx_2 * o : ou + idx_2 * o;
      const double* tensor_2 = at_x_2 ? oxx : ouu;
      int64_t L2 = at_x_2 ? n : m;
      double* tensor; int64_t L;
      if (at_x_1) { if (at_x_2) { tensor = oxx; L = n; } else { tensor = oux; L = m; } }
      else { tensor = ouu; L = m; }
      *in_var_2 = eps;
      integrate_x(f->p, x, dx, x1);
      for (int64_t k = 0; k < m; ++k) u1[k] = u[k] + du[k];
      fn_eval(f, x1, u1, f1);
      if (lie) difference_x(f->p, f0, f1, df);               /* difference_out (problem.hpp:268) */
      else for (int64_t k = 0; k < o; ++k) df[k] = f1[k] - f0[k];
      for (int64_t k = 0; k < o; ++k) df[k] -= eps * f_col_1[k];
      for (int64_t k = 0; k < o; ++k) df[k] -= eps * f_col_2[k];
      for (int64_t k = 0; k < o; ++k) df[k] *= 2;
      for (int64_t k = 0; k < o; ++k) {
        double val = 0.5 * (df[k] / eps2 - tensor_1[TIDX(k, idx_1, idx_1, o, L1)] - tensor_2[TIDX(k, idx_2, idx_2, o, L2)]);
        tensor[TIDX(k, idx_2, idx_1, o, L)] = val;
        if (at_x_1 == at_x_2) tensor[TIDX(k, idx_1, idx_2, o, L)] = val;
      }
      *in_var_2 = 0;
    }
    *in_var_1 = 0;
  }
  free(f1); free(df); free(x1); free(u1); free(dx); free(du);
}

/* finite_diff_hessian_compute::second_order_deriv, problem.hpp:300-337; fd_mode 0 (tensors left at
 * zero: the Gauss-Newton variant of SURVEY.md 8d) is an addition of the build, not of the reference */
static void fd_second_order(const fd_fn* f, const double* x, const double* u, double* oxx, double* oux, double* ouu,
                            double* ox, double* ou, double* out) {
  int64_t o = f->o, n = orc_ndx(f->p), m = orc_nu(f->p);
  if (f->p->fd_mode == 2) fd_second_order_2(f, x, u, oxx, oux, ouu, ox, ou, out);
  else if (f->p->fd_mode == 1) fd_second_order_1(f, x, u, oxx, oux, ouu, ox, ou, out);
  else {
    fn_first_order(f, x, u, ox, ou, out);
    memset(oxx, 0, sizeof(double) * (size_t)(o * n * n));
    memset(oux, 0, sizeof(double) * (size_t)(o * m * n));
    memset(ouu, 0, sizeof(double) * (size_t)(o * m * m));
  }
}

/* problem_t::compute_derivatives, problem.hpp:956-998 (without the print-only self check :999-1139) */
void orc_compute_derivatives(const orc_problem* p, const double* xs, const double* us, orc_derivs* d) {
  int64_t T = p->T, n = orc_ndx(p), m = orc_nu(p), nx = orc_nx(p);
  memset(d->lfx, 0, sizeof(double) * (size_t)n);
  memset(d->lfxx, 0, sizeof(double) * (size_t)(n * n));
  int64_t E = 0;
  for (int64_t t = 0; t < T; ++t) {
    const double* x = xs + t * nx;
    const double* u = us + t * m;
    int64_t e = p->ne[t];
    memset(d->lx + t * n, 0, sizeof(double) * (size_t)n);
    memset(d->lxx + t * n * n, 0, sizeof(double) * (size_t)(n * n));
    memset(d->lux + t * m * n, 0, sizeof(double) * (size_t)(m * n));
    for (int64_t i = 0; i < m; ++i) d->lu[t * m + i] = p->c * u[i];
    memset(d->luu + t * m * m, 0, sizeof(double) * (size_t)(m * m));
    for (int64_t i = 0; i < m; ++i) d->luu[t * m * m + i + i * m] = 1.0 * p->c;
    fd_fn ff = {p, 0, t, n};
    fd_second_order(&ff, x, u, d->fxx + t * n * n * n, d->fux + t * n * m * n, d->fuu + t * n * m * m,
                    d->fx + t * n * n, d->fu + t * n * m, d->f_val + t * nx);
    if (p->eq_kind != ORC_EQ_NONE && e > 0) {
      fd_fn fe = {p, 1, t, e};
      fd_second_order(&fe, x, u, d->eq_xx + E * n * n, d->eq_ux + E * m * n, d->eq_uu + E * m * m,
                      d->eq_x + E * n, d->eq_u + E * m, d->eq_val + E);
    }
    E += e;
  }
}

/* ddp_solver_t::make_trajectory, ddp.hpp:392-415 */
void orc_rollout(const orc_problem* p, const double* x0, const double* us, double* xs) {
  int64_t nx = orc_nx(p), m = orc_nu(p);
  memcpy(xs, x0, sizeof(double) * (size_t)nx);
  for (int64_t t = 0; t < p->T; ++t) orc_eval_f(p, xs + t * nx, us + t * m, xs + (t + 1) * nx);
}

/* ------------------------------------------------------------------------------------------ */
/* cost_seq_aug: ddp.hpp:699-735                                                               */
/* ------------------------------------------------------------------------------------------ */
void orc_cost_seq_aug(const orc_problem* p, const double* xs, const double* us, const orc_affine* mults, double mu,
                      double* costs) {
  int64_t T = p->T, n = orc_ndx(p), m = orc_nu(p), nx = orc_nx(p);
  int64_t E = 0;
  int64_t emax = 0;
  for (int64_t t = 0; t < T; ++t) if (p->ne[t] > emax) emax = p->ne[t];
  double* ce = dalloc(emax); double* pe = dalloc(emax); double* dxv = dalloc(n);
  for (int64_t t = 0; t < T; ++t) {
    const double* x = xs + t * nx;
    const double* u = us + t * m;
    int64_t e = p->ne[t];
    double un = 0;
    for (int64_t i = 0; i < m; ++i) un += u[i] * u[i];
    double l = 0.5 * p->c * un;  /* problem_t::l, problem.hpp:937-942 */
    double dot = 0, sq = 0;
    if (e > 0) {
      orc_eval_eq(p, t, x, u, ce);
      /* affine proxy operator(), mat_seq_common.hpp:105-115 */
      difference_x(p, mults->origin + t * nx, x, dxv);               /* x (-) origin */
      for (int64_t i = 0; i < e; ++i) pe[i] = mults->val[E + i];
      gemv_n_add(e, n, mults->jac + E * n, e, dxv, pe);
      for (int64_t i = 0; i < e; ++i) { dot += pe[i] * ce[i]; sq += ce[i] * ce[i]; }
    }
    costs[t] = l + dot + (mu / 2) * sq;
    E += e;
  }
  costs[T] = 0.0;  /* problem_t::lf, problem.hpp:932-936 */
  free(ce); free(pe); free(dxv);
}

/* ------------------------------------------------------------------------------------------ */
/* backward_pass: ddp_bwd.ipp:9-155                                                            */
/* ------------------------------------------------------------------------------------------ */
int64_t orc_backward(int64_t T, int64_t n, int64_t m, int64_t nx, const int64_t* ne, const orc_derivs* d,
                     const double* xs, const orc_affine* mults, double* reg_io, double* mu_io, orc_affine* fb,
                     double* Vx_trace, double* Vxx_trace, int heap_like, int64_t max_restarts) {
  double regularization = *reg_io, mu = *mu_io;
  int64_t emax = 0, Etot = 0;
  for (int64_t t = 0; t < T; ++t) { if (ne[t] > emax) emax = ne[t]; Etot += ne[t]; }
  int success = 0;
  int64_t restarts = 0;

  /* pre-allocated workspaces ("best-effort CPU"); the heap_like variant re-allocates per step
   * exactly where the reference's .eval() calls do (ddp_bwd.ipp:27-28,46-47,61-83) */
  double* V_xx = dalloc(n * n); double* V_x = dalloc(n);
  double* W_Q_x = dalloc(n); double* W_Q_u = dalloc(m); double* W_Q_xx = dalloc(n * n);
  double* W_Q_uu = dalloc(m * m); double* W_Q_ux = dalloc(m * n);
  double* W_tmp = dalloc(emax); double* W_tmp2 = dalloc(emax * n);
  double* W_A = dalloc(n * n); double* W_fact = dalloc(m * m);

  while (!success) {
    if (max_restarts >= 0 && restarts > max_restarts) { restarts = -1; break; }
    memcpy(V_xx, d->lfxx, sizeof(double) * (size_t)(n * n));   /* :27 */
    memcpy(V_x, d->lfx, sizeof(double) * (size_t)n);            /* :28 */
    int64_t E = Etot;
    int failed = 0;
    for (int64_t t = T - 1; t >= 0; --t) {                      /* reverse zip :31-39 */
      int64_t e = ne[t];
      E -= e;
      const double* lx = d->lx + t * n;   const double* lu = d->lu + t * m;
      const double* lxx = d->lxx + t * n * n; const double* lux = d->lux + t * m * n; const double* luu = d->luu + t * m * m;
      const double* fx = d->fx + t * n * n; const double* fu = d->fu + t * n * m;
      const double* fxx = d->fxx + t * n * n * n; const double* fux = d->fux + t * n * m * n; const double* fuu = d->fuu + t * n * m * m;
      const double* eqv = d->eq_val + E; const double* eqx = d->eq_x + E * n; const double* equ = d->eq_u + E * m;
      const double* eqxx = d->eq_xx + E * n * n; const double* equx = d->eq_ux + E * m * n; const double* equu = d->eq_uu + E * m * m;
      const double* pe = mults->val + E;
      const double* pe_x = mults->jac + E * n;

      double *Q_x, *Q_u, *Q_xx, *Q_uu, *Q_ux, *tmp, *tmp2, *A, *fact;
      if (heap_like) {
        Q_x = dalloc(n); Q_u = dalloc(m); Q_xx = dalloc(n * n); Q_uu = dalloc(m * m); Q_ux = dalloc(m * n);
        tmp = dalloc(e); tmp2 = dalloc(e * n); A = dalloc(n * n); fact = dalloc(m * m);
      } else {
        Q_x = W_Q_x; Q_u = W_Q_u; Q_xx = W_Q_xx; Q_uu = W_Q_uu; Q_ux = W_Q_ux;
        tmp = W_tmp; tmp2 = W_tmp2; A = W_A; fact = W_fact;
      }

      for (int64_t i = 0; i < e; ++i) tmp[i] = pe[i] + mu * eqv[i];                 /* :46 */
      for (int64_t i = 0; i < e * n; ++i) tmp2[i] = pe_x[i] + mu * eqx[i];          /* :47 */

      memcpy(Q_x, lx, sizeof(double) * (size_t)n);                                   /* :61 */
      gemv_t_add(n, n, fx, n, V_x, Q_x);                                              /* :62 */
      gemv_t_add(n, e, eqx, e, tmp, Q_x);                                             /* :63 */
      gemv_t_add(n, e, pe_x, e, eqv, Q_x);                                            /* :64 */

      memcpy(Q_u, lu, sizeof(double) * (size_t)m);                                   /* :66 */
      gemv_t_add(m, n, fu, n, V_x, Q_u);                                              /* :67 */
      gemv_t_add(m, e, equ, e, tmp, Q_u);                                             /* :68 */

      memcpy(Q_xx, lxx, sizeof(double) * (size_t)(n * n));                           /* :70 */
      memset(A, 0, sizeof(double) * (size_t)(n * n));
      gemm_tn_add(n, n, n, fx, n, V_xx, n, A, n);          /* (f.x^T V_xx), Eigen evaluates left to right */
      gemm_nn_add(n, n, n, A, n, fx, n, Q_xx, n);                                     /* :71 */
      gemm_tn_add(n, n, e, eqx, e, tmp2, e, Q_xx, n);                                 /* :72 */
      gemm_tn_add(n, n, e, pe_x, e, eqx, e, Q_xx, n);                                 /* :73 */
      contract_add_outdim(e, n, n, eqxx, tmp, Q_xx);                                  /* :74 */
      contract_add_outdim(n, n, n, fxx, V_x, Q_xx);                                   /* :75 */

      memcpy(Q_uu, luu, sizeof(double) * (size_t)(m * m));                           /* :77 */
      memset(A, 0, sizeof(double) * (size_t)(m * n));
      gemm_tn_add(m, n, n, fu, n, V_xx, n, A, m);
      gemm_nn_add(m, m, n, A, m, fu, n, Q_uu, m);                                     /* :78 */
      {
        /* (eq.u^T eq.u) * mu, :79 */
        for (int64_t j = 0; j < m; ++j)
          for (int64_t i = 0; i < m; ++i) {
            double s = 0;
            for (int64_t l = 0; l < e; ++l) s += equ[l + i * e] * equ[l + j * e];
            Q_uu[i + j * m] += s * mu;
          }
      }
      contract_add_outdim(e, m, m, equu, tmp, Q_uu);                                  /* :80 */
      contract_add_outdim(n, m, m, fuu, V_x, Q_uu);                                   /* :81 */

      memcpy(Q_ux, lux, sizeof(double) * (size_t)(m * n));                           /* :83 */
      gemm_nn_add(m, n, n, A, m, fx, n, Q_ux, m);                                     /* :84 (A = f.u^T V_xx) */
      gemm_tn_add(m, n, e, equ, e, tmp2, e, Q_ux, m);                                 /* :85 */
      contract_add_outdim(e, m, n, equx, tmp, Q_ux);                                  /* :86 */
      contract_add_outdim(n, m, n, fux, V_x, Q_ux);                                   /* :87 */

      memcpy(fact, Q_uu, sizeof(double) * (size_t)(m * m));                          /* :104 */
      for (int64_t i = 0; i < m; ++i) fact[i + i * m] += regularization;
      if (llt_lower(m, fact, m) >= 0) {                                               /* :105 */
        if (regularization < mu) regularization = mu;                                 /* :106-108 */
        mu *= 2;                                                                      /* :109 */
        regularization *= 2;                                                          /* :110 */
        failed = 1;
        if (heap_like) { free(Q_x); free(Q_u); free(Q_xx); free(Q_uu); free(Q_ux); free(tmp); free(tmp2); free(A); free(fact); }
        break;                                                                        /* :131 */
      }

      memcpy(fb->origin + t * nx, xs + t * nx, sizeof(double) * (size_t)nx);         /* :134 */
      double* k = fb->val + t * m;
      double* K = fb->jac + t * m * n;
      for (int64_t i = 0; i < m; ++i) k[i] = -Q_u[i];                                 /* :135 */
      llt_solve(m, fact, m, 1, k, m);
      for (int64_t i = 0; i < m * n; ++i) K[i] = -Q_ux[i];                            /* :136 */
      llt_solve(m, fact, m, n, K, m);

      memcpy(V_x, Q_x, sizeof(double) * (size_t)n);                                  /* :142 */
      gemv_t_add(n, m, Q_ux, m, k, V_x);                                              /* :143 */
      memcpy(V_xx, Q_xx, sizeof(double) * (size_t)(n * n));                          /* :145 */
      gemm_tn_add(n, n, m, Q_ux, m, K, m, V_xx, n);                                   /* :146 */

      if (Vx_trace) memcpy(Vx_trace + t * n, V_x, sizeof(double) * (size_t)n);
      if (Vxx_trace) memcpy(Vxx_trace + t * n * n, V_xx, sizeof(double) * (size_t)(n * n));

      if (heap_like) { free(Q_x); free(Q_u); free(Q_xx); free(Q_uu); free(Q_ux); free(tmp); free(tmp2); free(A); free(fact); }
      if (t == 0) success = 1;                                                        /* :149-151 */
    }
    if (failed) ++restarts;
  }
  free(V_xx); free(V_x); free(W_Q_x); free(W_Q_u); free(W_Q_xx); free(W_Q_uu); free(W_Q_ux);
  free(W_tmp); free(W_tmp2); free(W_A); free(W_fact);
  *reg_io = regularization;
  *mu_io = mu;
  return restarts;
}

/* ------------------------------------------------------------------------------------------ */
/* forward_pass: ddp_fwd.ipp:9-67                                                              */
/* ------------------------------------------------------------------------------------------ */
static void fwd_rollout(const orc_problem* p, double step, double* xs_new, double* us_new, const double* xs_old,
                        const double* us_old, const orc_affine* fb) {
  int64_t T = p->T, n = orc_ndx(p), m = orc_nu(p), nx = orc_nx(p);
  double* tmp = dalloc(n);
  for (int64_t t = 0; t < T; ++t) {
    const double* xo = xs_old + t * nx;
    double* xn = xs_new + t * nx;
    double* un = us_new + t * m;
    difference_x(p, xo, xn, tmp);                                                    /* :45 difference(out, old, new) */
    for (int64_t i = 0; i < m; ++i) un[i] = us_old[t * m + i] + step * fb->val[t * m + i]; /* :47-48 */
    {
      const double* K = fb->jac + t * m * n;
      for (int64_t i = 0; i < m; ++i) {
        double s = 0;
        for (int64_t l = 0; l < n; ++l) s += K[i + l * m] * tmp[l];
        un[i] += s;                                                                   /* :49 */
      }
    }
    orc_eval_f(p, xn, un, xs_new + (t + 1) * nx);                                     /* :50 */
  }
  free(tmp);
}

double orc_forward_alpha(const orc_problem* p, double step, double* xs_new, double* us_new, const double* xs_old,
                         const double* us_old, const orc_affine* mults, const orc_affine* fb, double mu) {
  int64_t T = p->T;
  double* co = dalloc(T + 1); double* cn = dalloc(T + 1);
  orc_cost_seq_aug(p, xs_old, us_old, mults, mu, co);
  fwd_rollout(p, step, xs_new, us_new, xs_old, us_old, fb);
  orc_cost_seq_aug(p, xs_new, us_new, mults, mu, cn);
  double s = 0;
  for (int64_t t = 0; t <= T; ++t) s += cn[t] - co[t];                                /* :56 */
  free(co); free(cn);
  return s;
}

double orc_forward(const orc_problem* p, double* xs_new, double* us_new, const double* xs_old, const double* us_old,
                   const orc_affine* mults, const orc_affine* fb, double mu, int64_t* n_evals) {
  int64_t T = p->T;
  double* co = dalloc(T + 1); double* cn = dalloc(T + 1);
  orc_cost_seq_aug(p, xs_old, us_old, mults, mu, co);                                 /* :24-26 */
  double step = 1;
  int success = 0;
  int64_t evals = 0;
  while (!success) {
    if (step < 1e-10) break;                                                          /* :35-37 */
    fwd_rollout(p, step, xs_new, us_new, xs_old, us_old, fb);
    ++evals;
    orc_cost_seq_aug(p, xs_new, us_new, mults, mu, cn);                               /* :54 */
    double s = 0;
    for (int64_t t = 0; t <= T; ++t) s += cn[t] - co[t];
    if (s <= 0) success = 1; else step *= 0.5;                                        /* :56-60 */
  }
  if (n_evals) *n_evals = evals;
  free(co); free(cn);
  return step;
}

/* ------------------------------------------------------------------------------------------ */
/* outer loop pieces: ddp.hpp:516-523, 576-627, 642-696, 745-842; mat_seq_common.hpp:62-89     */
/* ------------------------------------------------------------------------------------------ */
void orc_update_origin(const orc_problem* p, orc_affine* a, const int64_t* rows, const double* xs_new) {
  int64_t T = p->T, n = orc_ndx(p), nx = orc_nx(p);
  double* tmp = dalloc(n);
  int64_t R = 0;
  for (int64_t t = 0; t < T; ++t) {
    int64_t r = rows[t];
    difference_x(p, a->origin + t * nx, xs_new + t * nx, tmp);
    gemv_n_add(r, n, a->jac + R * n, r, tmp, a->val + R);
    /* jac = jac * d_difference_dfinish(origin, x_new) (mat_seq_common.hpp:80-86, problem.hpp:414-439): the identity on a
     * vector space; with a free-flyer root its leading 6 x 6 block is Jlog6 */
    if (m_ff(&p->model) && r > 0) {
      int nv = p->model.nv;
      double* Dq = dalloc((int64_t)nv * nv); double* nj = dzalloc(r * 6);
      orc_d_difference_dq_finish(&p->model, a->origin + t * nx, xs_new + t * nx, Dq);
      double* jac = a->jac + R * n;
      for (int64_t c = 0; c < 6; ++c)
        for (int64_t i = 0; i < r; ++i) {
          double s_ = 0;
          for (int64_t l = 0; l < 6; ++l) s_ += jac[i + l * r] * Dq[l + c * nv];
          nj[i + c * r] = s_;
        }
      memcpy(jac, nj, sizeof(double) * (size_t)(r * 6));
      free(Dq); free(nj);
    }
    memcpy(a->origin + t * nx, xs_new + t * nx, sizeof(double) * (size_t)nx);
    R += r;
  }
  free(tmp);
}

double orc_optimality_constr(const orc_problem* p, const orc_derivs* d) {
  double r = 0;
  int64_t E = 0;
  for (int64_t t = 0; t < p->T; ++t) {
    double s = 0;
    for (int64_t i = 0; i < p->ne[t]; ++i) s += d->eq_val[E + i] * d->eq_val[E + i];
    s = sqrt(s);
    if (s > r) r = s;
    E += p->ne[t];
  }
  return r;
}

double orc_optimality_obj(const orc_problem* p, const double* xs, const orc_affine* mults, double mu, const orc_derivs* d) {
  int64_t T = p->T, n = orc_ndx(p), m = orc_nu(p), nx = orc_nx(p);
  int64_t emax = 0, E = 0;
  for (int64_t t = 0; t < T; ++t) { if (p->ne[t] > emax) emax = p->ne[t]; E += p->ne[t]; }
  double* adj = dalloc(n); double* adj2 = dalloc(n); double* pe = dalloc(emax); double* lu = dalloc(m); double* dxv = dalloc(n);
  memcpy(adj, d->lfx, sizeof(double) * (size_t)n);
  double retval = 0;
  for (int64_t t = T - 1; t >= 0; --t) {
    int64_t e = p->ne[t];
    E -= e;
    const double* eqv = d->eq_val + E; const double* eqx = d->eq_x + E * n; const double* equ = d->eq_u + E * m;
    const double* fx = d->fx + t * n * n; const double* fu = d->fu + t * n * m;
    difference_x(p, mults->origin + t * nx, xs + t * nx, dxv);
    for (int64_t i = 0; i < e; ++i) pe[i] = mults->val[E + i];
    gemv_n_add(e, n, mults->jac + E * n, e, dxv, pe);
    memcpy(lu, d->lu + t * m, sizeof(double) * (size_t)m);
    gemv_t_add(m, e, equ, e, pe, lu);
    for (int64_t j = 0; j < m; ++j) {
      double s = 0;
      for (int64_t i = 0; i < e; ++i) s += mu * eqv[i] * equ[i + j * e];
      lu[j] += s;
    }
    gemv_t_add(m, n, fu, n, adj, lu);
    double nr = 0;
    for (int64_t j = 0; j < m; ++j) nr += lu[j] * lu[j];
    nr = sqrt(nr);
    if (nr > retval) retval = nr;
    memset(adj2, 0, sizeof(double) * (size_t)n);
    gemv_t_add(n, n, fx, n, adj, adj2);
    for (int64_t j = 0; j < n; ++j) adj2[j] += d->lx[t * n + j];
    for (int64_t j = 0; j < n; ++j) {
      double s = 0;
      for (int64_t i = 0; i < e; ++i) s += mu * eqv[i] * eqx[i + j * e];
      adj2[j] += s;
    }
    gemv_t_add(n, e, eqx, e, pe, adj2);
    gemv_t_add(n, e, mults->jac + E * n, e, eqv, adj2);
    memcpy(adj, adj2, sizeof(double) * (size_t)n);
  }
  free(adj); free(adj2); free(pe); free(lu); free(dxv);
  return retval;
}

static void derivs_alloc(const orc_problem* p, orc_derivs* d) {
  int64_t T = p->T, n = orc_ndx(p), m = orc_nu(p), nx = orc_nx(p), E = orc_ne_total(p);
  d->lfx = dzalloc(n); d->lfxx = dzalloc(n * n);
  d->lx = dzalloc(T * n); d->lu = dzalloc(T * m); d->lxx = dzalloc(T * n * n); d->lux = dzalloc(T * m * n); d->luu = dzalloc(T * m * m);
  d->f_val = dzalloc(T * nx); d->fx = dzalloc(T * n * n); d->fu = dzalloc(T * n * m);
  d->fxx = dzalloc(T * n * n * n); d->fux = dzalloc(T * n * m * n); d->fuu = dzalloc(T * n * m * m);
  d->eq_val = dzalloc(E); d->eq_x = dzalloc(E * n); d->eq_u = dzalloc(E * m);
  d->eq_xx = dzalloc(E * n * n); d->eq_ux = dzalloc(E * m * n); d->eq_uu = dzalloc(E * m * m);
}
static void derivs_free(orc_derivs* d) {
  free(d->lfx); free(d->lfxx); free(d->lx); free(d->lu); free(d->lxx); free(d->lux); free(d->luu);
  free(d->f_val); free(d->fx); free(d->fu); free(d->fxx); free(d->fux); free(d->fuu);
  free(d->eq_val); free(d->eq_x); free(d->eq_u); free(d->eq_xx); free(d->eq_ux); free(d->eq_uu);
}

void orc_solve(const orc_problem* p, int64_t max_iterations, double threshold, double mu, double reg, double w,
               double nn, const double* mult_jac_seed, double* xs, double* us, orc_affine* fb, orc_solve_log* log) {
  int64_t T = p->T, n = orc_ndx(p), m = orc_nu(p), nx = orc_nx(p), E = orc_ne_total(p);
  orc_derivs d;
  derivs_alloc(p, &d);                                                                /* ddp.hpp:750 */
  double* xs_new = dalloc((T + 1) * nx); double* us_new = dalloc(T * m);
  memcpy(xs_new, xs, sizeof(double) * (size_t)((T + 1) * nx));                        /* :752 clone */
  memcpy(us_new, us, sizeof(double) * (size_t)(T * m));
  orc_affine mults = {dalloc(T * nx), dzalloc(E), dalloc(E * n)};                     /* :759-764 */
  memcpy(mults.jac, mult_jac_seed, sizeof(double) * (size_t)(E * n));
  memcpy(mults.origin, xs, sizeof(double) * (size_t)(T * nx));
  int64_t* urows = (int64_t*)malloc(sizeof(int64_t) * (size_t)T);
  for (int64_t t = 0; t < T; ++t) urows[t] = m;

  orc_compute_derivatives(p, xs, us, &d);                                             /* :768 */
  double mu_b = mu, reg_b = reg;
  orc_backward(T, n, m, nx, p->ne, &d, xs, &mults, &reg_b, &mu_b, fb, NULL, NULL, 0, 1000);   /* :769 */
  mu = mu_b;                                                                          /* :771 (reg is NOT taken) */
  double step = orc_forward(p, xs_new, us_new, xs, us, &mults, fb, mu, NULL);         /* :772 */

  memset(log, 0, sizeof(*log));
  int64_t iter = 0;
  for (; iter < max_iterations; ++iter) {
    /* update_derivatives, ddp.hpp:642-696 */
    orc_compute_derivatives(p, xs, us, &d);
    orc_update_origin(p, &mults, p->ne, xs);
    orc_update_origin(p, fb, urows, xs);
    double opt_obj = orc_optimality_obj(p, xs, &mults, mu, &d);
    double opt_constr = orc_optimality_constr(p, &d);
    log->opt_obj = opt_obj; log->opt_constr = opt_constr;
    if (opt_constr < threshold && opt_obj < threshold) { log->result = 1; break; }    /* :673-675 */
    if (opt_obj < w) {
      if (opt_constr < nn) {
        int64_t Eo = 0;
        for (int64_t t = 0; t < T; ++t) {                                             /* :680-688 */
          int64_t e = p->ne[t];
          const double* eqv = d.eq_val + Eo; const double* eqx = d.eq_x + Eo * n; const double* equ = d.eq_u + Eo * m;
          const double* k = fb->val + t * m; const double* K = fb->jac + t * m * n;
          for (int64_t i = 0; i < e; ++i) {
            double s = eqv[i];
            for (int64_t l = 0; l < m; ++l) s += equ[i + l * e] * k[l];
            mults.val[Eo + i] += mu * s;
          }
          for (int64_t j = 0; j < n; ++j)
            for (int64_t i = 0; i < e; ++i) {
              double s = eqx[i + j * e];
              for (int64_t l = 0; l < m; ++l) s += equ[i + l * e] * K[l + j * m];
              mults.jac[Eo * n + i + j * e] += mu * s;
            }
          Eo += e;
        }
        double oo = orc_optimality_obj(p, xs, &mults, mu, &d);                        /* :795-797 */
        nn = oo / pow(mu, 0.1);
        w /= pow(mu, 1.0);
      } else {
        mu *= 10;                                                                     /* :791 */
      }
    }
    mu_b = mu; reg_b = reg;
    orc_backward(T, n, m, nx, p->ne, &d, xs, &mults, &reg_b, &mu_b, fb, NULL, NULL, 0, 1000);  /* :804 */
    mu = mu_b; reg = reg_b;                                                           /* :805-806 */
    step = orc_forward(p, xs_new, us_new, xs, us, &mults, fb, mu, NULL);              /* :817 */
    if (step >= 0.5) {                                                                /* :819-824 */
      reg /= 2;
      if (reg < 1e-5) reg = 0;
    }
    {                                                                                 /* :826 swap */
      double* sx = dalloc((T + 1) * nx); double* su = dalloc(T * m);
      memcpy(sx, xs, sizeof(double) * (size_t)((T + 1) * nx)); memcpy(su, us, sizeof(double) * (size_t)(T * m));
      memcpy(xs, xs_new, sizeof(double) * (size_t)((T + 1) * nx)); memcpy(us, us_new, sizeof(double) * (size_t)(T * m));
      memcpy(xs_new, sx, sizeof(double) * (size_t)((T + 1) * nx)); memcpy(us_new, su, sizeof(double) * (size_t)(T * m));
      free(sx); free(su);
    }
  }
  log->iterations = iter; log->mu = mu; log->reg = reg; log->w = w; log->n = nn; log->last_step = step;
  free(mults.origin); free(mults.val); free(mults.jac); free(urows); free(xs_new); free(us_new);
  derivs_free(&d);
}
