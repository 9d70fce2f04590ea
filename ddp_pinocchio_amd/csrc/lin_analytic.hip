// lin_analytic.hip -- analytic first order (problem.hpp:463-503 with d_dynamics_aba, pinocchio_model.ipp:359-400) and
// mode-1 second order (finite_diff_hessian_compute::second_order_deriv_1, problem.hpp:67-150: forward differences of the
// analytic jacobians, eps = sqrt(DBL_EPSILON)) of large tree models (the Talos-like tree), gfx950.
//
// An "evaluation" is one first_order_deriv at a point (x (+) eps e_p, u): p = 0 the trajectory point, p = 1 .. nv a
// q direction, p = nv+1 .. 2nv a v direction.  The u directions need no evaluation: M^-1 depends on q alone, so
// fu(x, u + eps e) == fu(x, u) bit for bit and f_uu = (fu_ - fu) / eps is exactly zero, as in the reference; for the
// same reason the v slabs of f_ux are exactly zero and the v directions reuse the base point's M^-1.
// Three kernels per slice of (instance, t) pairs, communicating through an HBM workspace (L2 / MALL sized slices):
//   ana_eval_kernel   one wave per evaluation: forward dynamics (rbd::aba_tree_coop), then the world-frame recursion of
//                     rbd_deriv.h -> T = [d tau/dq | d tau/dv] (nv x 2nv) and, for p <= nv, the joint-space inertia M
//   ana_minv_kernel   one wave per (pair, configuration): Cholesky of M and M^-1, rows in registers
//   ana_out_kernel    one wave per evaluation: -M^-1 T on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), then f_x, f_u
//                     (p = 0) or the tensor slabs f_xx(:,:,p-1), f_ux(:,:,p-1) = (jacobian' - jacobian) / eps (p > 0)
//   ana_eq_kernel     (constrained problems) one wave per (pair, direction): the constraint chain
//                     constraint_advance_time_t::first_order_deriv (problem.hpp:569-605) on the analytic jacobians, at the
//                     trajectory point (eq_val, eq_x, eq_u) and -- mode 1 -- at x (+) eps e_i, u + eps e_i, differenced into
//                     eq_xx, eq_ux, eq_uu (problem.hpp:67-150 with Fn = the constraint chain, :611-620)
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include "internal.h"
#include "lin_common.h"
#include "rbd.h"
#include "rbd_deriv.h"

namespace {

constexpr int AW = 64;     // lanes per evaluation (one wave)
constexpr int ANA_F = 1, ANA_EQ = 2;

struct AnaParams {
  LinParams lp;
  double* Tws;      // [slice pairs][2nv+1][nv][2nv]  T, row-major (the row of a joint is contiguous)
  double* Mws;      // [slice pairs][nv+1][nv][nv]    M, then M^-1 in place (symmetric)
  int64_t bt0;      // first (instance, t) pair of the slice
  int32_t nbt;      // pairs in the slice
  int32_t stage;    // 0: first order only (p = 0), 1: the perturbed points (p = 1 .. 2nv)
  double* Fws;      // [slice pairs][2nv][nv][2nv] the v rows of f_x at the perturbed points (kept for the constraint chain), or null
  int32_t write_f;  // stage 1: form the f_xx / f_ux slabs (0: this pass only serves the constraint tensors)
  int32_t pad_;
  const double* accel;   // stage 1: [pair][2nv][nv] accelerations of the perturbed points, formed by the static first-order kernels
                         // (lin_static.hip, level 6) -- or null: every evaluation runs its own forward dynamics
};

// ---- kernel A -----------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(AW) void ana_eval_kernel(AnaParams ap) {
  const LinParams& p = ap.lp;
  const DevModel& m = *p.model;
  const int N = m.nv, W2 = 2 * N;
  // stage 1 re-evaluates the trajectory point (p = 0) as well: the v directions use its M^-1, and the workspace slice
  // may have been recycled since stage 0
  const int P = ap.stage == 0 ? 1 : 2 * N + 1;            // evaluations per pair in this launch
  const int64_t e = blockIdx.x;
  const int64_t sbt = e / P;                              // pair within the slice
  const int pp = (int)(e % P);                            // perturbation index
  const int64_t bt = ap.bt0 + sbt;
  const int64_t T = p.d.T;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const int lane = threadIdx.x;

  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int R1 = 78 * NJ > rbd::ABA_LDS_SLOTS * NJ ? 78 * NJ : rbd::ABA_LDS_SLOTS * NJ;
  double* s_R1 = lds;                 // ABA state, then per joint Ic[36] | Bc[36] | ofc[6], then T (nv x 2nv)
  double* s_W = s_R1 + R1;            // per joint oR[9] | op[3] | J[6] | ov[6] | oa[6]
  double* s_P = s_W + 30 * NJ;        // per joint y | z | u | g | Fq | Fv
  double* s_q = s_P + 36 * NJ;
  double* s_v = s_q + NJ;
  double* s_tau = s_v + NJ;
  double* s_a = s_tau + NJ;

  {
    const double* xs = p.x + ((int64_t)b * (T + 1) + t) * (2 * N);
    const double* us = p.u + ((int64_t)b * T + t) * N;
    const double eps = sqrt(DBL_EPSILON);
    for (int i = lane; i < N; i += AW) {
      double qi = xs[i], vi = xs[N + i];
      if (pp >= 1 && pp - 1 == i) qi = qi + eps;          // integrate_x, problem.hpp:107,117
      if (pp >= 1 && pp - 1 == N + i) vi = vi + eps;
      s_q[i] = qi; s_v[i] = vi; s_tau[i] = us[i];
    }
  }
  __syncthreads();
  if (ap.accel != nullptr && pp >= 1) {
    const double* __restrict__ ag = ap.accel + ((int64_t)bt * W2 + (pp - 1)) * N;
    for (int i = lane; i < N; i += AW) s_a[i] = ag[i];
    __syncthreads();
  } else {
    rbd::aba_tree_coop<NJ, 1, AW>(m, s_q, s_v, s_tau, s_a, s_R1, 0, lane, true);   // ends with a barrier
  }
  // world-frame recursion, root -> leaves, one lane per joint of a level
  for (int L = 0; L < m.n_levels; ++L) {
    const int idx = m.lvl_start[L] + lane;
    if (idx < m.lvl_start[L + 1]) {
      const int i = m.lvl_joint[idx];
      const int par = m.parent[i];
      double oR[9], op[3], J[6], ov[6], oa[6], oRp[9], opp[3];
      if (par >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) oRp[k] = s_W[30 * par + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) opp[k] = s_W[30 * par + 9 + k];
      }
      rbdd::world_placement(m, i, s_q[i], par >= 0 ? oRp : nullptr, par >= 0 ? opp : nullptr, oR, op);
      rbdd::world_axis(m, i, oR, op, J);
      double vJ[6], t6[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vJ[k] = J[k] * s_v[i];
#pragma unroll
      for (int k = 0; k < 6; ++k) ov[k] = (par >= 0 ? s_W[30 * par + 18 + k] : 0.0) + vJ[k];
      rbd::crm(ov, vJ, t6);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double apk = par >= 0 ? s_W[30 * par + 24 + k] : (k < 3 ? 0.0 : -m.gravity[k - 3]);
        oa[k] = apk + J[k] * s_a[i] + t6[k];
      }
      double* w = s_W + 30 * i;
#pragma unroll
      for (int k = 0; k < 9; ++k) w[k] = oR[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) w[9 + k] = op[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) { w[12 + k] = J[k]; w[18 + k] = ov[k]; w[24 + k] = oa[k]; }
    }
    __syncthreads();
  }
  // per body: world inertia, force, bias matrix (the ABA state in s_R1 is dead)
  for (int i = lane; i < N; i += AW) {
    const double* w = s_W + 30 * i;
    double oR[9], op[3], ov[6], oa[6], I6[36], B[36], h[6], Ioa[6], vxh[6];
#pragma unroll
    for (int k = 0; k < 9; ++k) oR[k] = w[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) op[k] = w[9 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { ov[k] = w[18 + k]; oa[k] = w[24 + k]; }
    rbdd::world_inertia(m.I6[i], oR, op, I6);
    rbdd::m6v(I6, ov, h);
    rbdd::m6v(I6, oa, Ioa);
    rbd::crf(ov, h, vxh);
    rbdd::bias_matrix(I6, ov, h, B);
    double* o = s_R1 + 78 * i;
#pragma unroll
    for (int k = 0; k < 36; ++k) { o[k] = I6[k]; o[36 + k] = B[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) o[72 + k] = Ioa[k] + vxh[k];
  }
  __syncthreads();
  // composite sums leaves -> root: lane = entry; a lane only ever touches its own entries, children precede parents
  for (int i = N - 1; i >= 1; --i) {
    const int par = m.parent[i];
    if (par < 0) continue;
    for (int k = lane; k < 78; k += AW) s_R1[78 * par + k] += s_R1[78 * i + k];
  }
  __syncthreads();
  for (int i = lane; i < N; i += AW) {
    const double* w = s_W + 30 * i;
    const double* c = s_R1 + 78 * i;
    double J[6], ov[6], oa[6], Ic[36], Bc[36], ofc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { J[k] = w[12 + k]; ov[k] = w[18 + k]; oa[k] = w[24 + k]; ofc[k] = c[72 + k]; }
#pragma unroll
    for (int k = 0; k < 36; ++k) { Ic[k] = c[k]; Bc[k] = c[36 + k]; }
    double y[6], z[6], u[6], g[6], t1[6], t2[6], t3[6];
    rbdd::m6v(Ic, J, y);
    rbdd::m6tv(Bc, J, z);
    rbd::crm(J, ov, u);
    rbd::crm(u, ov, t1);
    rbd::crm(J, oa, t2);
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] = t1[k] - t2[k];
    double* o = s_P + 36 * i;
#pragma unroll
    for (int k = 0; k < 6; ++k) { o[k] = y[k]; o[6 + k] = z[k]; o[12 + k] = u[k]; o[18 + k] = g[k]; }
    rbd::crf(J, ofc, t1);
    rbdd::m6v(Bc, u, t2);
    rbdd::m6v(Ic, g, t3);
#pragma unroll
    for (int k = 0; k < 6; ++k) o[24 + k] = t1[k] - t2[k] + t3[k];
    rbdd::m6v(Bc, J, t1);
    rbdd::m6v(Ic, u, t2);
#pragma unroll
    for (int k = 0; k < 6; ++k) o[30 + k] = t1[k] - 2.0 * t2[k];
  }
  __syncthreads();
  // T (row-major nv x 2nv: [d tau/dq | d tau/dv]) in LDS over the dead composite region, M straight to the workspace
  double* s_T = s_R1;
  for (int k = lane; k < N * W2; k += AW) s_T[k] = 0.0;
  __syncthreads();
  const bool want_M = pp <= N;
  double* Mo = ap.Mws + (sbt * (N + 1) + (want_M ? pp : 0)) * (int64_t)N * N;
  if (want_M)
    for (int k = lane; k < N * N; k += AW) Mo[k] = 0.0;
  __syncthreads();
  for (int j = lane; j < N; j += AW) {
    const double* Pj = s_P + 36 * j;
    for (int i = j; i >= 0; i = m.parent[i]) {            // i in path(j): column j, row i
      const double* Ji = s_W + 30 * i + 12;
      double sq = 0, sv = 0, sm = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { sq += Ji[k] * Pj[24 + k]; sv += Ji[k] * Pj[30 + k]; sm += Ji[k] * Pj[k]; }
      s_T[i * W2 + j] = sq;
      s_T[i * W2 + N + j] = sv;
      if (want_M) { Mo[i + (int64_t)j * N] = sm; Mo[j + (int64_t)i * N] = sm; }
    }
    for (int a = m.parent[j]; a >= 0; a = m.parent[a]) {  // a proper ancestor of j: row j, column a
      const double* Pa = s_P + 36 * a;
      const double* Ja = s_W + 30 * a + 12;
      double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { s1 += Pj[6 + k] * Pa[12 + k]; s2 += Pj[k] * Pa[18 + k]; s3 += Pj[6 + k] * Ja[k]; s4 += Pj[k] * Pa[12 + k]; }
      s_T[j * W2 + a] = -s1 + s2;
      s_T[j * W2 + N + a] = s3 - 2.0 * s4;
    }
  }
  __syncthreads();
  double* To = ap.Tws + (sbt * (2 * N + 1) + pp) * (int64_t)N * W2;
  for (int k = lane; k < N * W2; k += AW) To[k] = s_T[k];
}

// ---- kernel B: M -> M^-1 (in place in the workspace); lane = row / right-hand side ------------------------------------
template <int NJ>
__global__ __launch_bounds__(AW) void ana_minv_kernel(AnaParams ap) {
  const int N = (int)ap.lp.d.nv;
  const int C = ap.stage == 0 ? 1 : N + 1;                // configurations per pair in this launch
  const int64_t e = blockIdx.x;
  const int64_t sbt = e / C;
  const int c = (int)(e % C);
  double* Mg = ap.Mws + (sbt * (N + 1) + c) * (int64_t)N * N;
  constexpr int LD = NJ | 1;
  __shared__ double sL[NJ * LD];                          // L, column-major with an odd leading dimension
  __shared__ double sD[NJ];                               // 1 / L_kk
  const int r = threadIdx.x;
  const bool live = r < N;
  double a[NJ];                                           // row r of the lower triangle
#pragma unroll
  for (int j = 0; j < NJ; ++j) a[j] = (live && j <= r && j < N) ? Mg[r + (int64_t)j * N] : 0.0;
  // right-looking Cholesky: column k is final after step k and goes to LDS for the others to read
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    if (k < N) {
      if (r == k) { const double dk = sqrt(a[k]); a[k] = dk; sL[k + k * LD] = dk; sD[k] = 1.0 / dk; }
      __syncthreads();
      const double dinv = sD[k];
      if (live && r > k) { a[k] = a[k] * dinv; sL[r + k * LD] = a[k]; }
      __syncthreads();
      const double lrk = a[k];
#pragma unroll
      for (int j = k + 1; j < NJ; ++j)
        if (j < N) a[j] = (live && r >= j) ? a[j] - lrk * sL[j + k * LD] : a[j];
    }
  }
  __syncthreads();
  // lane r solves L L^T x = e_r: forward then backward substitution, L read as LDS broadcasts
  double x[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) x[i] = (i == r) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    if (i < N) {
      double s = x[i];
#pragma unroll
      for (int l = 0; l < i; ++l) s -= sL[i + l * LD] * x[l];
      x[i] = s * sD[i];
    }
  }
#pragma unroll
  for (int i = NJ - 1; i >= 0; --i) {
    if (i < N) {
      double s = x[i];
#pragma unroll
      for (int l = i + 1; l < NJ; ++l)
        if (l < N) s -= sL[l + i * LD] * x[l];
      x[i] = s * sD[i];
    }
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      if (i < N) Mg[i + (int64_t)r * N] = x[i];           // column r of M^-1
  }
}

// ---- kernel C: R = -M^-1 T on the matrix cores, then the outputs --------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NJ>
__global__ __launch_bounds__(AW) void ana_out_kernel(AnaParams ap) {
  const LinParams& p = ap.lp;
  const DevModel& m = *p.model;
  const int N = m.nv, W2 = 2 * N, n = 2 * N;
  const int P = ap.stage == 0 ? 1 : 2 * N + 1;
  const int64_t e = blockIdx.x;
  const int64_t sbt = e / P;
  const int pp = (int)(e % P);
  if (ap.stage == 1 && pp == 0) return;                   // the trajectory point's jacobians are stage 0's
  const int64_t bt = ap.bt0 + sbt;
  const int lane = threadIdx.x;
  const int l15 = lane & 15, l4 = lane >> 4;
  const double* Tt = ap.Tws + (sbt * (2 * N + 1) + pp) * (int64_t)N * W2;            // T row-major: T(i, j) at i * 2nv + j
  const double* Mi = ap.Mws + (sbt * (N + 1) + (pp <= N ? pp : 0)) * (int64_t)N * N; // M^-1 of this evaluation's configuration
  const double dt = m.dt;
  const double eps = sqrt(DBL_EPSILON);

  double* fx = p.fx + bt * (int64_t)n * n;
  double* fu = p.fu + bt * (int64_t)n * N;
  double* slab_xx = pp > 0 ? p.fxx + (bt * n + (pp - 1)) * (int64_t)n * n : nullptr;   // f_xx(:, :, p-1): n x n
  double* slab_ux = pp > 0 ? p.fux + (bt * n + (pp - 1)) * (int64_t)n * N : nullptr;   // f_ux(:, :, p-1): n x nv

  // D'(j, r) = sum_l T(l, j) Minv(l, r): A(row = j, k = l) = T(l, j), B(k = l, col = r) = Minv(l, r); tiles 16 x 16, k by 4.
  // Result register q of lane: D'(row = 16 jt + l4 + 4 q, col = 16 rt + l15): 16 consecutive r per quarter wave = 128 B
  const int JT = (W2 + 15) / 16, RT = (N + 15) / 16, KS = (N + 3) / 4;
  for (int jt = 0; jt < JT; ++jt) {
    for (int rt = 0; rt < RT; ++rt) {
      const int ja = 16 * jt + l15, rb = 16 * rt + l15;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int s = 0; s < KS; ++s) {
        const int l = 4 * s + l4;
        const double av = (ja < W2 && l < N) ? Tt[(int64_t)l * W2 + ja] : 0.0;
        const double bv = (rb < N && l < N) ? Mi[l + (int64_t)rb * N] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 16 * jt + l4 + 4 * q;               // column of the jacobian block (0 .. 2nv-1: q then v directions)
        const int r = rb;                                 // row (joint)
        if (j >= W2 || r >= N) continue;
        // first_order_deriv, problem.hpp:499-501: fx_bot = dt * d qdd/dx (+ I on the v block)
        double val = (-acc[q]) * dt;
        if (j >= N && j - N == r) val = val + 1.0;
        const int64_t off = (N + r) + (int64_t)j * n;
        if (pp == 0) fx[off] = val;
        else {
          if (ap.write_f) slab_xx[off] = (val - fx[off]) / eps;        // problem.hpp:128-137
          if (ap.Fws) ap.Fws[(sbt * (2 * N) + (pp - 1)) * (int64_t)N * n + r + (int64_t)j * N] = val;
        }
      }
    }
  }
  if (pp > 0 && !ap.write_f) return;
  // the rows of q+ = q + dt v: constants (problem.hpp:487-490), so their differences are exact zeros
  for (int k = lane; k < N * n; k += AW) {
    const int i = k % N, j = k / N;
    const int64_t off = i + (int64_t)j * n;
    if (pp == 0) fx[off] = (j == i) ? 1.0 : ((j == N + i) ? 1.0 * dt : 0.0);
    else slab_xx[off] = 0.0;
  }
  // f_u = [0; dt M^-1] (problem.hpp:493,502)
  for (int k = lane; k < n * N; k += AW) {
    const int i = k % n, j = k / n;
    const int64_t off = i + (int64_t)j * n;
    if (pp == 0) fu[off] = i < N ? 0.0 : Mi[(i - N) + (int64_t)j * N] * dt;
    else if (pp <= N) slab_ux[off] = i < N ? 0.0 : (Mi[(i - N) + (int64_t)j * N] * dt - fu[off]) / eps;   // problem.hpp:138-140
    else slab_ux[off] = 0.0;                              // a v direction: the same M^-1, fu_ == fu
  }
}

// ---- kernel D: the constraint chain on the analytic jacobians -----------------------------------------------------------
// constraint_advance_time_t::first_order_deriv (problem.hpp:569-605) wrapped K = eq_advance times around a base constraint
// that depends on q alone (config_constraint_t :744-864, spatial_constraint_t :631-742: jacobian C = [C_q | 0]):
//   x_{k+1} = f(x_k, u)  (the SAME u at every look-ahead step, :563-567),  eq = c(x_K),
//   eq_x = C f_x(x_{K-1}) .. f_x(x_0),  eq_u = C f_x(x_{K-1}) .. f_x(x_1) f_u(x_0)                         (:603-604)
// On a vector-space model the q rows of an analytic f_x are the constants [I | dt I] (:487-490) and C has no v columns, so
// the first product of the chain, C f_x(x_{K-1}) = [C_q | dt C_q], holds whatever x_{K-1} is: every term the full product
// adds on top is an exact zero (0 * finite), and the sum the reference forms is reproduced bit for bit without a jacobian at
// the look-ahead state.  With K <= 2 -- what every reference driver uses -- that is the whole chain (K > 2 would need the
// full f_x at x_1 .. x_{K-2}: refused at ddp_hip_create).
// dir = 0: the trajectory point -> eq_val, eq_x, eq_u.  dir = 1 .. n + m (mode 1, problem.hpp:105-147): the point
// x (+) eps e_i / u + eps e_i -> eq_xx(:,:,i), eq_ux(:,:,i) resp. eq_uu(:,:,i) = (jacobian' - jacobian) / eps.  f_x at a
// perturbed x comes from ana_out (Fws), f_u there is dt M^-1(q') (Mws); a u direction leaves f_u as it is, bit for bit.
template <int NJ>
__global__ __launch_bounds__(AW) void ana_eq_kernel(AnaParams ap) {
  const LinParams& p = ap.lp;
  const DevModel& m = *p.model;
  const int N = m.nv, n = 2 * N, K = m.eq_advance;
  const int P = ap.stage == 0 ? 1 : 3 * N;                // directions per pair in this launch
  const int64_t blk = blockIdx.x;
  const int64_t sbt = blk / P;
  const int dir = ap.stage == 0 ? 0 : 1 + (int)(blk % P); // 1 .. 2N: x directions (ana_eval's numbering), 2N+1 .. 3N: u directions
  const int64_t bt = ap.bt0 + sbt;
  const int64_t T = p.d.T;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const int e = (int)p.ne[t];
  if (e == 0) return;
  const int lane = threadIdx.x;
  const int64_t Eo = p.Epre[t], Eb = (int64_t)b * p.d.Etot + Eo;
  const double dt = m.dt;
  const double eps = sqrt(DBL_EPSILON);

  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* s_st = lds;                                   // ABA state
  double* s_q = s_st + rbd::ABA_LDS_SLOTS * NJ;
  double* s_v = s_q + NJ;
  double* s_tau = s_v + NJ;
  double* s_a = s_tau + NJ;
  double* s_C = s_a + NJ;                               // C_q: e x nv, column-major

  {
    const double* xs = p.x + ((int64_t)b * (T + 1) + t) * n;
    const double* us = p.u + ((int64_t)b * T + t) * N;
    for (int i = lane; i < N; i += AW) {
      double qi = xs[i], vi = xs[N + i], ui = us[i];
      if (dir >= 1 && dir - 1 == i) qi = qi + eps;        // integrate_x / integrate_u, problem.hpp:107,117-118
      if (dir >= 1 && dir - 1 == N + i) vi = vi + eps;
      if (dir >= 1 && dir - 1 == 2 * N + i) ui = ui + eps;
      s_q[i] = qi; s_v[i] = vi; s_tau[i] = ui;
    }
  }
  __syncthreads();
  for (int k = 0; k < K; ++k) {                           // dynamics_t::eval_to, problem.hpp:441-461
    rbd::aba_tree_coop<NJ, 1, AW>(m, s_q, s_v, s_tau, s_a, s_st, 0, lane, true);   // ends with a barrier
    for (int i = lane; i < N; i += AW) {
      const double vo = dt * s_v[i];
      s_q[i] = s_q[i] + vo;
      s_v[i] = s_v[i] + s_a[i] * dt;
    }
    __syncthreads();
  }
  const double* target = p.target + Eo;
  double val = 0.0;                                       // lane i < e: row i of the constraint value
  if (m.eq_kind == DDP_HIP_EQ_CONFIG) {
    for (int k = lane; k < e * N; k += AW) s_C[k] = (k % e == k / e) ? 1.0 : 0.0;   // d_difference_dq_finish = I (problem.hpp:834-842)
    if (lane < e) val = s_q[lane] - target[lane];
  } else {
    if (lane == 0) {
      double pos[3], J[3 * NJ];
      rbd::frame_position<NJ>(m, s_q, pos, J);
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < e; ++i) s_C[i + j * e] = J[i + 3 * j];
      for (int i = 0; i < e; ++i) s_a[i] = pos[i] - target[i];
    }
    __syncthreads();
    if (lane < e) val = s_a[lane];
  }
  __syncthreads();
  auto C1 = [&](int i, int l) -> double {                 // C f_x(x_{K-1}) = [C_q | dt C_q]; K == 1: C itself
    if (l < N) return s_C[i + l * e];
    return K >= 2 ? s_C[i + (l - N) * e] * dt : 0.0;
  };
  const double* fx = p.fx + bt * (int64_t)n * n;
  const double* fu = p.fu + bt * (int64_t)n * N;
  if (dir == 0) {
    if (lane < e) p.eq_val[Eb + lane] = val;
    for (int idx = lane; idx < e * n; idx += AW) {
      const int i = idx % e, j = idx / e;
      double sacc = 0.0;
      for (int l = 0; l < n; ++l) sacc += C1(i, l) * fx[l + (int64_t)j * n];
      p.eq_x[Eb * n + idx] = sacc;
    }
    for (int idx = lane; idx < e * N; idx += AW) {
      const int i = idx % e, j = idx / e;
      double sacc = 0.0;
      for (int l = 0; l < n; ++l) sacc += C1(i, l) * fu[l + (int64_t)j * n];
      p.eq_u[Eb * N + idx] = sacc;
    }
    return;
  }
  const double* ox = p.eq_x + Eb * n;
  const double* ou = p.eq_u + Eb * N;
  if (dir <= 2 * N) {
    const int idx3 = dir - 1;                             // the right index of the slab
    const double* Fv = ap.Fws + (sbt * (2 * N) + idx3) * (int64_t)N * n;                       // v rows of f_x at the perturbed point
    const double* Mi = ap.Mws + (sbt * (N + 1) + (dir <= N ? dir : 0)) * (int64_t)N * N;       // M^-1 of its configuration
    double* sxx = p.eq_xx + Eb * n * n + (int64_t)idx3 * e * n;
    double* sux = p.eq_ux + Eb * N * n + (int64_t)idx3 * e * N;
    for (int idx = lane; idx < e * n; idx += AW) {
      const int i = idx % e, j = idx / e;
      // the q rows of f_x are the constants [I | dt I] (problem.hpp:487-490): one non-zero term, the others exact zeros
      double sacc = j < N ? C1(i, j) * 1.0 : C1(i, j - N) * (1.0 * dt);
      for (int r = 0; r < N; ++r) sacc += C1(i, N + r) * Fv[r + (int64_t)j * N];
      sxx[idx] = (sacc - ox[idx]) / eps;                  // problem.hpp:128-134
    }
    for (int idx = lane; idx < e * N; idx += AW) {
      const int i = idx % e, j = idx / e;
      double sacc = 0.0;                                  // the q rows of f_u are zero (problem.hpp:493)
      for (int r = 0; r < N; ++r) sacc += C1(i, N + r) * (Mi[r + (int64_t)j * N] * dt);
      sux[idx] = (sacc - ou[idx]) / eps;                  // problem.hpp:135-137
    }
  } else {
    const int idx3 = dir - 1 - 2 * N;
    double* suu = p.eq_uu + Eb * N * N + (int64_t)idx3 * e * N;
    for (int idx = lane; idx < e * N; idx += AW) {
      const int i = idx % e, j = idx / e;
      double sacc = 0.0;
      for (int l = 0; l < n; ++l) sacc += C1(i, l) * fu[l + (int64_t)j * n];   // f_u(x, u + eps e) == f_u(x, u), bit for bit
      suu[idx] = (sacc - ou[idx]) / eps;                  // problem.hpp:141-145
    }
  }
}

template <int NJ>
size_t eq_lds_bytes(const Dims& d) { return sizeof(double) * (size_t)(rbd::ABA_LDS_SLOTS * NJ + 4 * NJ + d.emax * NJ); }

// flags: ANA_F the dynamics' own outputs (stage 0: f_x, f_u; stage 1: f_xx, f_ux, f_uu), ANA_EQ the constraint chain's
// (stage 0: eq_val, eq_x, eq_u from the resident f_x, f_u; stage 1: eq_xx, eq_ux, eq_uu)
template <int NJ>
int launch_t(ddp_hip_ctx* ctx, const LinParams& p, int stage, int flags) {
  const Dims& d = ctx->d;
  const int64_t BT = d.batch * d.T;
  const int N = (int)d.nv;
  constexpr int R1 = 78 * NJ > rbd::ABA_LDS_SLOTS * NJ ? 78 * NJ : rbd::ABA_LDS_SLOTS * NJ;
  const size_t lds = sizeof(double) * (size_t)(R1 + 30 * NJ + 36 * NJ + 4 * NJ);
  const bool do_f = (flags & ANA_F) != 0;
  const bool do_eq = (flags & ANA_EQ) != 0 && d.Etot > 0;
  AnaParams ap{};
  ap.lp = p;
  ap.Tws = ctx->ana_T;
  ap.Mws = ctx->ana_M;
  ap.stage = stage;
  ap.write_f = do_f ? 1 : 0;
  ap.Fws = (stage == 1 && do_eq) ? ctx->ana_F : nullptr;
  if (stage == 1 && do_eq && !ctx->ana_F) return DDP_HIP_E_UNSUPPORTED;
  if (stage == 1 && do_f) {
    // f_uu is exactly zero (see the header of this file)
    HIP_TRY(hipMemsetAsync(p.fuu, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_FUU].size * d.batch), ctx->stream));
  }
  if (stage == 1 && ctx->ana_A && ctx->lin_static && p.qcache) {
    // the forward dynamics of the 2 nv perturbed points: the static first-order kernels evaluate exactly these points
    // (x + sqrt(eps_mach) e_k) chain-wise from the base point's cache, ~25x cheaper than one cooperative ABA per point
    LinParams pa = p;
    pa.accel_out = ctx->ana_A;
    const int rc_ = lin_static_launch(ctx, pa, 6);
    if (rc_ != DDP_HIP_OK) return rc_;
    ap.accel = ctx->ana_A;
  }
  const int P = stage == 0 ? 1 : 2 * N + 1, C = stage == 0 ? 1 : N + 1;
  if (stage == 0 && !do_f) {
    // the base point of the constraint chain alone: it reads the resident f_x, f_u
    if (do_eq) {
      ap.bt0 = 0; ap.nbt = (int32_t)BT;
      hipLaunchKernelGGL((ana_eq_kernel<NJ>), dim3((unsigned)BT), dim3(AW), eq_lds_bytes<NJ>(d), ctx->stream, ap);
    }
    HIP_TRY(hipGetLastError());
    return DDP_HIP_OK;
  }
  for (int64_t bt0 = 0; bt0 < BT; bt0 += ctx->ana_nbt) {
    const int64_t nb = BT - bt0 < ctx->ana_nbt ? BT - bt0 : ctx->ana_nbt;
    ap.bt0 = bt0;
    ap.nbt = (int32_t)nb;
    hipLaunchKernelGGL((ana_eval_kernel<NJ>), dim3((unsigned)(nb * P)), dim3(AW), lds, ctx->stream, ap);
    hipLaunchKernelGGL((ana_minv_kernel<NJ>), dim3((unsigned)(nb * C)), dim3(AW), 0, ctx->stream, ap);
    hipLaunchKernelGGL((ana_out_kernel<NJ>), dim3((unsigned)(nb * P)), dim3(AW), 0, ctx->stream, ap);
    if (do_eq)
      hipLaunchKernelGGL((ana_eq_kernel<NJ>), dim3((unsigned)(nb * (stage == 0 ? 1 : 3 * N))), dim3(AW), eq_lds_bytes<NJ>(d), ctx->stream, ap);
  }
  HIP_TRY(hipGetLastError());
  return DDP_HIP_OK;
}

}  // namespace

int lin_analytic_setup(ddp_hip_ctx* ctx) {
  const Dims& d = ctx->d;
  if (ctx->model_h.kind != DDP_HIP_MODEL_TREE || ctx->model_h.first_order_fd || d.nv <= 6) return DDP_HIP_OK;
  if (ctx->model_h.max_level_width > AW) return DDP_HIP_E_UNSUPPORTED;
  const int64_t BT = d.batch * d.T;
  int64_t slice = 256;
  if (const char* ev = getenv("DDP_HIP_ANA_BT")) { const int v = atoi(ev); if (v >= 1 && v <= 65536) slice = v; }   // tuning knob
  ctx->ana_nbt = BT < slice ? BT : slice;
  const int64_t N = d.nv;
  // the stage-0 and stage-1 launches of one linearisation share the workspace: the base point keeps slot 0 of every pair
  HIP_TRY(hipMalloc(&ctx->ana_T, sizeof(double) * (size_t)(ctx->ana_nbt * (2 * N + 1) * N * 2 * N)));
  HIP_TRY(hipMalloc(&ctx->ana_M, sizeof(double) * (size_t)(ctx->ana_nbt * (N + 1) * N * N)));
  if (ctx->lin_static && ctx->lin_ws && ctx->model_h.fd_mode == 1 && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS))
    HIP_TRY(hipMalloc(&ctx->ana_A, sizeof(double) * (size_t)(BT * 2 * N * N)));
  if (d.Etot > 0) {
    // the constraint chain on analytic jacobians (ana_eq_kernel): K <= 2 look-ahead steps, see the kernel's header
    if (ctx->model_h.eq_advance < 1 || ctx->model_h.eq_advance > 2) return DDP_HIP_E_UNSUPPORTED;
    if (ctx->model_h.fd_mode == 1 && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS))
      HIP_TRY(hipMalloc(&ctx->ana_F, sizeof(double) * (size_t)(ctx->ana_nbt * 2 * N * N * 2 * N)));
    const size_t l38 = eq_lds_bytes<38>(d), l64 = eq_lds_bytes<64>(d);
    if (l38 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ana_eq_kernel<38>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l38));
    if (d.nv > 38 && l64 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ana_eq_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l64));
  }
  constexpr int NJ = 64;
  constexpr int R1 = 78 * NJ > rbd::ABA_LDS_SLOTS * NJ ? 78 * NJ : rbd::ABA_LDS_SLOTS * NJ;
  if (d.nv > 38) {
    const size_t lds = sizeof(double) * (size_t)(R1 + 30 * NJ + 36 * NJ + 4 * NJ);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ana_eval_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  return DDP_HIP_OK;
}

void lin_analytic_teardown(ddp_hip_ctx* ctx) {
  if (ctx->ana_T) (void)hipFree(ctx->ana_T);
  if (ctx->ana_M) (void)hipFree(ctx->ana_M);
  if (ctx->ana_F) (void)hipFree(ctx->ana_F);
  if (ctx->ana_A) (void)hipFree(ctx->ana_A);
}

int lin_analytic_launch(ddp_hip_ctx* ctx, const LinParams& p, int stage, int flags) {
  if (!ctx->ana_T || !ctx->ana_M) return DDP_HIP_E_UNSUPPORTED;
  if (stage == 1 && !p.has_tensors) return DDP_HIP_OK;
  if (ctx->d.nv <= 38) return launch_t<38>(ctx, p, stage, flags);
  return launch_t<64>(ctx, p, stage, flags);
}
