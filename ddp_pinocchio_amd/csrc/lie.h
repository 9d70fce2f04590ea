// lie.h -- Lie-group configurations on the device: a free-flyer root joint (SE(3); q = [p(3), quaternion x y z w],
// v = [linear(3), angular(3)] in the body frame, Pinocchio's JointModelFreeFlyer) next to 1-DoF joints.  Replaces what
// model_t::integrate / difference / d_difference_dq_finish (pinocchio_model.ipp:222-321) delegate to Pinocchio (absent):
// closed forms of exp / log on SE(3) and of the Jacobian of the logarithm (Barfoot, State Estimation for Robotics,
// eqs. 7.85-7.86 for the coupling block).  State-level helpers follow dynamics_t::integrate_x / difference_out
// (problem.hpp:395-412).  With a free-flyer root nq = nv + 1; joint j >= 1 uses q[j + 6], v[j + 5].
#pragma once

#include "internal.h"

namespace lie {

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void quat_to_R(const double* qt, double* R) {   // x y z w, unit; row-major, world = R * body
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* c) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
  c[0] = aw * bx + ax * bw + ay * bz - az * by;
  c[1] = aw * by - ax * bz + ay * bw + az * bx;
  c[2] = aw * bz + ax * by - ay * bx + az * bw;
  c[3] = aw * bw - ax * bx - ay * by - az * bz;
}
// coefficients in t = |w|: b = (1 - cos t)/t^2, c = (t - sin t)/t^3, d = (1 - (t/2) cot(t/2))/t^2.  The closed forms cancel
// near 0 (relative error ~ 6 eps / t^2 for c, 12 eps / t^2 for d: 2e-8 at t = 1e-4, where round 2 switched over), so the
// Taylor series runs up to t^2 < 0.04 (six terms: truncation < 1e-16) and the closed forms -- written on the half angle, which
// removes the 1 - cos t cancellation -- take over where they are good to 1e-13.  oracle/ddp_oracle.c:so3_coeffs is the same code.
__device__ __forceinline__ void so3_coeffs(double t2, double& b, double& c, double& d) {
  if (t2 < 0.04) {
    b = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 + t2 * (-1.0 / 479001600)))));
    c = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 + t2 * (-1.0 / 6227020800.0)))));
    d = 1.0 / 12 + t2 * (1.0 / 720 + t2 * (1.0 / 30240 + t2 * (1.0 / 1209600 + t2 * (1.0 / 47900160 + t2 * (691.0 / 1307674368000.0)))));
  } else {
    const double t = sqrt(t2);
    double st, ct, sh, ch;
    sincos(t, &st, &ct);
    sincos(0.5 * t, &sh, &ch);
    b = 2.0 * sh * sh / t2; c = (t - st) / (t2 * t); d = (1 - 0.5 * t * ch / sh) / t2;
  }
}
// c4 = (1 - t^2/2 - cos t)/t^4, c6 = (t - sin t - t^3/6)/t^5 (Barfoot's Q block): cancellation ~ 24 eps / t^4 resp. 120 eps / t^4,
// so the series (eight terms) runs up to t^2 < 1
__device__ __forceinline__ void so3_coeffs_q(double t2, double& c4, double& c6) {
  if (t2 < 1.0) {
    c4 = -1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 + t2 * (-1.0 / 479001600 + t2 * (1.0 / 87178291200.0 +
         t2 * (-1.0 / 20922789888000.0 + t2 * (1.0 / 6402373705728000.0)))))));
    c6 = -1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 + t2 * (-1.0 / 6227020800.0 + t2 * (1.0 / 1307674368000.0 +
         t2 * (-1.0 / 355687428096000.0 + t2 * (1.0 / 121645100408832000.0)))))));
  } else {
    const double t = sqrt(t2);
    double st, ct;
    sincos(t, &st, &ct);
    c4 = (1 - t2 / 2 - ct) / (t2 * t2); c6 = (t - st - t2 * t / 6) / (t2 * t2 * t);
  }
}
__device__ __forceinline__ void quat_exp(const double* w, double* qt) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double k, cw;
  if (t2 < 1e-8) { k = 0.5 - t2 / 48; cw = 1 - t2 / 8 + t2 * t2 / 384; }
  else { const double t = sqrt(t2); double sh, ch; sincos(0.5 * t, &sh, &ch); k = sh / t; cw = ch; }
  qt[0] = k * w[0]; qt[1] = k * w[1]; qt[2] = k * w[2]; qt[3] = cw;
}
__device__ __forceinline__ void quat_log(const double* qin, double* w) {
  double qt[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (qt[3] < 0) { qt[0] = -qt[0]; qt[1] = -qt[1]; qt[2] = -qt[2]; qt[3] = -qt[3]; }
  const double n2 = qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2];
  double k;
  if (n2 < 1e-16) k = 2.0 / qt[3] * (1 - n2 / (3 * qt[3] * qt[3]));
  else { const double nn = sqrt(n2); k = 2 * atan2(nn, qt[3]) / nn; }
  w[0] = k * qt[0]; w[1] = k * qt[1]; w[2] = k * qt[2];
}
// y = (I + alpha [w]x + beta [w]x^2) x
__device__ __forceinline__ void so3_apply(const double* w, double alpha, double beta, const double* x, double* y) {
  double wx[3], wwx[3];
  cross3(w, x, wx);
  cross3(w, wx, wwx);
#pragma unroll
  for (int k = 0; k < 3; ++k) y[k] = x[k] + alpha * wx[k] + beta * wwx[k];
}
// q7' = q7 (+) nu, nu = (v, w) body twist
__device__ void se3_integrate(const double* q7, const double* nu, double* out7) {
  const double* v = nu; const double* w = nu + 3;
  double R[9], b, c, d, pe[3], qe[4], qn[4];
  so3_coeffs(w[0] * w[0] + w[1] * w[1] + w[2] * w[2], b, c, d);
  so3_apply(w, b, c, v, pe);
  quat_to_R(q7 + 3, R);
#pragma unroll
  for (int k = 0; k < 3; ++k) out7[k] = q7[k] + (R[3 * k] * pe[0] + R[3 * k + 1] * pe[1] + R[3 * k + 2] * pe[2]);
  quat_exp(w, qe);
  quat_mul(q7 + 3, qe, qn);
  if (qn[0] * q7[3] + qn[1] * q7[4] + qn[2] * q7[5] + qn[3] * q7[6] < 0) { qn[0] = -qn[0]; qn[1] = -qn[1]; qn[2] = -qn[2]; qn[3] = -qn[3]; }
  const double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) out7[3 + k] = qn[k] / nn;
}
// nu = q1 (-) q0 = log6(M0^-1 M1)
__device__ void se3_difference(const double* q0, const double* q1, double* nu) {
  double R0[9], dp[3], rp[3], qr[4], w[3], b, c, d;
  const double q0c[4] = {-q0[3], -q0[4], -q0[5], q0[6]};
  quat_mul(q0c, q1 + 3, qr);
  quat_log(qr, w);
  quat_to_R(q0 + 3, R0);
#pragma unroll
  for (int k = 0; k < 3; ++k) dp[k] = q1[k] - q0[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) rp[k] = R0[k] * dp[0] + R0[3 + k] * dp[1] + R0[6 + k] * dp[2];
  so3_coeffs(w[0] * w[0] + w[1] * w[1] + w[2] * w[2], b, c, d);
  so3_apply(w, -0.5, d, rp, nu);
  nu[3] = w[0]; nu[4] = w[1]; nu[5] = w[2];
}
__device__ __forceinline__ void skew(const double* a, double* S) {
  S[0] = 0; S[1] = -a[2]; S[2] = a[1]; S[3] = a[2]; S[4] = 0; S[5] = -a[0]; S[6] = -a[1]; S[7] = a[0]; S[8] = 0;
}
__device__ __forceinline__ void mm3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// Jlog6 at exp6(nu) (= d(q1 (-) q0)/dq1 in the tangent at q1, pinocchio dDifference ARG1); row-major 6 x 6, [linear; angular]
__device__ void se3_Jlog(const double* nu, double* J) {
  const double* v = nu; const double* w = nu + 3;
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double b, c, d;
  so3_coeffs(t2, b, c, d);
  double W[9], V[9], W2[9], Ji[9], Q[9], T1[9], T2[9];
  skew(w, W); skew(v, V);
  mm3(W, W, W2);
#pragma unroll
  for (int k = 0; k < 9; ++k) Ji[k] = 0.5 * W[k] + d * W2[k];
  Ji[0] += 1; Ji[4] += 1; Ji[8] += 1;
  double c4, c6;
  so3_coeffs_q(t2, c4, c6);
  const double c5 = 0.5 * (c4 - 3 * c6);
  double nW[9], nV[9], WV[9], VW[9], WVW[9], WWV[9], VWW[9], WVWW[9], WWVW[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { nW[k] = -W[k]; nV[k] = -V[k]; }
  mm3(nW, nV, WV); mm3(nV, nW, VW);
  mm3(WV, nW, WVW);
  mm3(nW, WV, WWV); mm3(VW, nW, VWW);
  mm3(WVW, nW, WVWW); mm3(nW, WVW, WWVW);
#pragma unroll
  for (int k = 0; k < 9; ++k)
    Q[k] = 0.5 * nV[k] + c * (WV[k] + VW[k] + WVW[k]) - c4 * (WWV[k] + VWW[k] - 3 * WVW[k]) - c5 * (WVWW[k] + WWVW[k]);
  mm3(Ji, Q, T1); mm3(T1, Ji, T2);
#pragma unroll
  for (int k = 0; k < 36; ++k) J[k] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { J[6 * i + j] = Ji[3 * i + j]; J[6 * (i + 3) + j + 3] = Ji[3 * i + j]; J[6 * i + j + 3] = -T2[3 * i + j]; }
}

// ---- state level (x = [q(nq); v(nv)], tangent dimension 2 nv) -----------------------------------------------------------
// x (+) eps e_idx in place: dynamics_t::integrate_x (problem.hpp:395-401) for a single tangent direction
__device__ __forceinline__ void perturb_x(const DevModel& m, double* x, int idx, double eps) {
  if (!m.ff) { x[idx] = x[idx] + eps; return; }
  const int nv = m.nv;
  if (idx >= nv) { x[m.nq + idx - nv] = x[m.nq + idx - nv] + eps; return; }
  if (idx >= 6) { x[idx + 1] = x[idx + 1] + eps; return; }
  double nu[6] = {0, 0, 0, 0, 0, 0}, q7[7];
  nu[idx] = eps;
  se3_integrate(x, nu, q7);
#pragma unroll
  for (int k = 0; k < 7; ++k) x[k] = q7[k];
}
// out(2 nv) = x1 (-) x0: difference_out for states (problem.hpp:403-412)
__device__ __forceinline__ void difference_x(const DevModel& m, const double* x0, const double* x1, double* out) {
  const int nv = m.nv;
  if (!m.ff) { for (int i = 0; i < 2 * nv; ++i) out[i] = x1[i] - x0[i]; return; }
  se3_difference(x0, x1, out);
  for (int i = 6; i < nv; ++i) out[i] = x1[i + 1] - x0[i + 1];
  for (int i = 0; i < nv; ++i) out[nv + i] = x1[m.nq + i] - x0[m.nq + i];
}

}  // namespace lie
