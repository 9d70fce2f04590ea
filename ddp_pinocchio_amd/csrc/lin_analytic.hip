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
#include <stdint.h>
#include <stdlib.h>
#include <utility>

#include "internal.h"
#include "lin_common.h"
#include "rbd.h"
#include "rbd_deriv.h"

namespace {

constexpr int AW = 64;     // lanes per evaluation (one wave)
constexpr int ANA_F = 1, ANA_EQ = 2, ANA_ACCEL = 4, ANA_EQ_NEXT = ANA_EQ << 4;

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
  double* M0;       // fused path: [B T][nv][nv] M^-1 at the trajectory points (stage 0 writes it, the v directions of stage 1 read it)
  int32_t m0_only;  // fused path, stage 0: form M0 alone (the pre-pass of a stage-1 launch)
  int32_t accel_base;  // stage 0: `accel` also holds the acceleration of the trajectory point itself (slot 3 nv): no forward dynamics in this launch
  int32_t pad4_;
  int32_t eq_inline;   // fused path, stage 1, config constraint: eq_xx / eq_ux slabs are written by the evaluation's own wave (no Fws / Mws, no ana_eq launch)
  int32_t pad3_;
  int32_t eq_no_aba;   // ana_eq_kernel: no evaluation of this launch runs forward dynamics (stage 1 with `accel`): no ABA state in LDS
  const double* accel;   // stage 1: [pair][3 nv: q, v, u directions][nv] accelerations of the perturbed points, formed by the static first-order kernels
                         // (lin_static.hip, level 6) -- or null: every evaluation runs its own forward dynamics
};

// ---- LDS layout of ana_eval_kernel (doubles) -----------------------------------------------------------------------------
//   evaluation:  R1 [78 NJ: ABA state, then per joint Ic 36 | Bc 36 | ofc 6] | W [30 NJ: oR 9 | op 3 | J 6 | ov 6 | oa 6, later u | g over
//                ov | oa] | q, v, tau, a [4 NJ] | parent, depth (int) [NJ]
//   assembly on: T [nv x 2nv] over the dead R1 records when it fits there (2 NJ <= 78), else behind the evaluation block;
//                the image of L, then M^-1 (fused path) behind T, resp. over the dead records
// (measured: the readlane form of the in-wave inverse 77 k cycles per wave, the LDS-image form 87 k -- the default is the former)
#ifndef DDP_ANA_INV_LDS
#define DDP_ANA_INV_LDS 0
#endif
constexpr bool INV_LDS = DDP_ANA_INV_LDS != 0;
template <int NJ> struct AnaLds {
  static constexpr int R1 = (78 > rbd::ABA_LDS_SLOTS ? 78 : rbd::ABA_LDS_SLOTS) * NJ;
  static constexpr int W = R1, Q = W + 30 * NJ, INTS = Q + 4 * NJ, EVAL = INTS + NJ;
  static constexpr bool T_OVER = 2 * NJ <= 78;
  static constexpr int T = T_OVER ? 0 : EVAL;
  static constexpr int X = T_OVER ? 2 * NJ * NJ : 0, X_SZ = INV_LDS ? NJ * (NJ | 1) : NJ * NJ;
  static constexpr int m2(int a, int b) { return a > b ? a : b; }
  static constexpr int TOTAL = m2(EVAL, m2(T + 2 * NJ * NJ, X + X_SZ));
};

// ---- in-wave helpers (fused path) -------------------------------------------------------------------------------------
// A wave that is alone on its SIMD pays the full latency of every LDS read it waits for one by one.  batch_fence closes a batch
// of reads (a compiler-level memory fence keeps later reads from being hoisted over it), value_fence pins a value so that the
// arithmetic on a batch is not sunk below the next one: one wait per batch and a bounded number of registers in flight.
__device__ __forceinline__ void batch_fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void value_fence(double& v) { asm volatile("" : "+v"(v)); }
// ana_eval_kernel's work-group is ONE wave (AW = 64), and the LDS executes a wave's accesses in program order: a value one lane
// wrote is there for any lane's later read without s_barrier.  What is left of a barrier is the compiler-level ordering -- and,
// unlike __syncthreads, no wait for the global reads the wave has in flight (the prefetched jacobians)
static_assert(AW == 64, "wave_sync assumes a one-wave work-group");
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); }
// f(integral_constant<int, 0>{}), f(integral_constant<int, 1>{}), ...: a loop whose counter is a template constant in the body
template <class F, int... Js>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Js...>) { (f(std::integral_constant<int, Js>{}), ...); }
// the value lane `src` (a constant) holds, as a wave-uniform scalar: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// M -> M^-1 with the wave's registers alone.  Lane r holds row r of the lower triangle of M in a[0 .. r] (rows r >= nv: the rows
// of an identity block, so that nothing below depends on nv).  Right-looking Cholesky M = L L^T, row r of L replacing row r of M;
// then lane r solves L L^T x = e_r.  Every L(i, l) another lane needs is one lane_bcast of the owner's register: the arithmetic
// is that of the split path's ana_minv_kernel, entry for entry.  (Entries above the diagonal are never read; they hold junk.)
// (the broadcasts of a chunk go out together, into scalar pairs of their own, then the chunk's arithmetic: a v_readlane_b32 pair
// followed at once by its consumer costs the hazard wait states every time)
// LB: the L(j, k) go through an LDS image of L instead (column-major, leading dimension NJ | 1; written by the owning lane, read
// as broadcasts, eight reads in flight per wait): an LDS read is not a VALU instruction, and a wave that is alone on its SIMD
// issues one FP64 VALU instruction per 8 clocks -- 2 readlanes + 1 FMA per entry against 1 FMA.  (Measured slower all the same: the
// waits on the batches cost more than the readlanes' issue slots; kept behind -DDDP_ANA_INV_LDS=1.)
constexpr int BC = 8;
template <bool LB, int K, int J0, int NJ, int... U>
__device__ __forceinline__ void chol_update_chunk(double (&a)[NJ], const double* sL, std::integer_sequence<int, U...>) {
  constexpr int LD = NJ | 1;
  const double l[] = {(LB ? sL[(J0 + U) + K * LD] : lane_bcast(a[K], J0 + U))...};   // L(j, k): lane j's a[k]
  ((a[J0 + U] = a[J0 + U] - a[K] * l[U]), ...);                                  // rows r >= j
  (value_fence(a[J0 + U]), ...);
  if constexpr (LB) batch_fence(); else __builtin_amdgcn_sched_barrier(0);
}
template <bool LB, int K, int J0, int CNT, int NJ>
__device__ __forceinline__ void chol_update(double (&a)[NJ], const double* sL) {
  if constexpr (CNT > 0) {
    constexpr int C = CNT < BC ? CNT : BC;
    chol_update_chunk<LB, K, J0>(a, sL, std::make_integer_sequence<int, C>{});
    chol_update<LB, K, J0 + C, CNT - C>(a, sL);
  }
}
// sx -= sum_u L(I, L0 + u) x[L0 + u] (FWD: L(I, l) is lane I's a[l]) resp. L(L0 + u, I) x[L0 + u] (lane L0 + u's a[I])
template <bool LB, bool FWD, int I, int L0, int NJ, int... U>
__device__ __forceinline__ void subst_chunk(const double (&a)[NJ], const double* sL, const double (&x)[NJ], double& sx, std::integer_sequence<int, U...>) {
  constexpr int LD = NJ | 1;
  const double l[] = {(LB ? (FWD ? sL[I + (L0 + U) * LD] : sL[(L0 + U) + I * LD]) : (FWD ? lane_bcast(a[L0 + U], I) : lane_bcast(a[I], L0 + U)))...};
  ((sx -= l[U] * x[L0 + U]), ...);
  value_fence(sx);
  if constexpr (LB) batch_fence(); else __builtin_amdgcn_sched_barrier(0);
}
template <bool LB, bool FWD, int I, int L0, int CNT, int NJ>
__device__ __forceinline__ void subst_range(const double (&a)[NJ], const double* sL, const double (&x)[NJ], double& sx) {
  if constexpr (CNT > 0) {
    constexpr int C = CNT < BC ? CNT : BC;
    subst_chunk<LB, FWD, I, L0>(a, sL, x, sx, std::make_integer_sequence<int, C>{});
    subst_range<LB, FWD, I, L0 + C, CNT - C>(a, sL, x, sx);
  }
}
template <bool LB, int K, int NJ>
__device__ __forceinline__ void chol_steps(double (&a)[NJ], double* sL, double& my_dinv, int r) {
  if constexpr (K < NJ) {
    constexpr int LD = NJ | 1;
    const double dk_own = sqrt(a[K]), di_own = 1.0 / dk_own;
    const double dk = lane_bcast(dk_own, K), dinv_k = lane_bcast(di_own, K);
    my_dinv = (r == K) ? di_own : my_dinv;
    a[K] = (r == K) ? dk : a[K] * dinv_k;
    if constexpr (LB) {
      if (r < NJ) sL[r + K * LD] = a[K];                  // column K of L (the entries above the diagonal: junk nobody reads)
      batch_fence();
    }
    chol_update<LB, K, K + 1, NJ - K - 1>(a, sL);
    chol_steps<LB, K + 1>(a, sL, my_dinv, r);
  }
}
template <bool LB, int I, int NJ>
__device__ __forceinline__ void fwd_rows(const double (&a)[NJ], const double* sL, double (&x)[NJ], double my_dinv) {
  if constexpr (I < NJ) {
    double sx = x[I];
    subst_range<LB, true, I, 0, I>(a, sL, x, sx);
    x[I] = sx * lane_bcast(my_dinv, I);
    value_fence(x[I]);
    fwd_rows<LB, I + 1>(a, sL, x, my_dinv);
  }
}
template <bool LB, int I, int NJ>
__device__ __forceinline__ void bwd_rows(const double (&a)[NJ], const double* sL, double (&x)[NJ], double my_dinv) {
  if constexpr (I >= 0) {
    double sx = x[I];
    subst_range<LB, false, I, I + 1, NJ - I - 1>(a, sL, x, sx);
    x[I] = sx * lane_bcast(my_dinv, I);
    value_fence(x[I]);
    bwd_rows<LB, I - 1>(a, sL, x, my_dinv);
  }
}
template <bool LB, int NJ>
__device__ __forceinline__ void wave_spd_inverse(double (&a)[NJ], double (&x)[NJ], int r, double* sL) {
  double my_dinv = 0.0;                                   // 1 / L(r, r)
  chol_steps<LB, 0>(a, sL, my_dinv, r);
  // (the substitutions broadcast the same L(i, l) the factorisation did; the fences keep the compiler from parking all nv^2 / 2
  // of them in spilled scalars to save the second and third v_readlane)
#pragma unroll
  for (int i = 0; i < NJ; ++i) value_fence(a[i]);
#pragma unroll
  for (int i = 0; i < NJ; ++i) x[i] = (i == r) ? 1.0 : 0.0;
  fwd_rows<LB, 0>(a, sL, x, my_dinv);
#pragma unroll
  for (int i = 0; i < NJ; ++i) value_fence(a[i]);
  bwd_rows<LB, NJ - 1>(a, sL, x, my_dinv);
  if constexpr (LB) batch_fence();                        // the image of L is dead: M^-1 goes over it
}

#ifdef DEV_ANA_CLOCKS   // development: cycle stamps of one wave per launch
#define CLK(k) do { if (lane == 0) clk[k] = __builtin_readcyclecounter(); } while (0)
#else
#define CLK(k) do { } while (0)
#endif
// ---- kernel A -----------------------------------------------------------------------------------------------------
template <int NJ, bool FUSED>
__global__ __launch_bounds__(AW) void ana_eval_kernel(AnaParams ap) {
  const LinParams& p = ap.lp;
  const DevModel& m = *p.model;
  const int N = m.nv, W2 = 2 * N;
  // split path: stage 1 re-evaluates the trajectory point (p = 0) as well -- the v directions use its M^-1, and the workspace
  // slice may have been recycled since stage 0.  Fused path: that M^-1 is resident (AnaParams::M0), stage 1 runs p = 1 .. 2nv
  const int P = ap.stage == 0 ? 1 : (FUSED ? 2 * N : 2 * N + 1);   // evaluations per pair in this launch
  const int64_t e = blockIdx.x;
  const int64_t sbt = e / P;                              // pair within the slice
  const int pp = ap.stage == 0 ? 0 : (int)(e % P) + (FUSED ? 1 : 0);   // perturbation index
  const int64_t bt = ap.bt0 + sbt;
  const int64_t T = p.d.T;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const int lane = threadIdx.x;

#ifdef DEV_ANA_CLOCKS
  unsigned long long clk[12] = {0};
#endif
  CLK(0);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  typedef AnaLds<NJ> LY;
  double* s_R1 = lds;                 // ABA state, then per joint Ic[36] | Bc[36] | ofc[6]
  double* s_W = lds + LY::W;          // per joint oR[9] | op[3] | J[6] | ov[6] | oa[6]; u | g replace ov | oa once they are spent
  double* s_q = lds + LY::Q;
  double* s_v = s_q + NJ;
  double* s_tau = s_v + NJ;
  double* s_a = s_tau + NJ;
  int* s_par = reinterpret_cast<int*>(s_a + NJ);   // parent of joint i

  // lane i owns joint i (nv <= 64): its constants, its state and its own placement stay in registers, so the model table is read
  // once, by all lanes at the same time, and the level loop below touches nothing but LDS
  const bool live = lane < N;
  const int ji = live ? lane : 0;
  const int par_i = m.parent[ji];
  const bool revolute = m.jtype[ji] == DDP_HIP_JOINT_REVOLUTE;
  double Rl[9], rl[3], ax[3], vi;
  {
    const double* xs = p.x + ((int64_t)b * (T + 1) + t) * (2 * N);
    const double* us = p.u + ((int64_t)b * T + t) * N;
    const double eps = sqrt(DBL_EPSILON);
    double qi = xs[ji];
    vi = xs[N + ji];
    if (pp >= 1 && pp - 1 == ji) qi = qi + eps;           // integrate_x, problem.hpp:107,117
    if (pp >= 1 && pp - 1 == N + ji) vi = vi + eps;
    if (live) { s_q[lane] = qi; s_v[lane] = vi; s_tau[lane] = us[lane]; s_par[lane] = par_i; }
    double E[9];
    rbd::joint_placement(m, ji, qi, E, rl);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) Rl[3 * k + l] = E[3 * l + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) ax[k] = m.axis[ji][k];
    if (live) {                                           // the joint's own placement, for its descendants' walks
      double* w = s_W + 30 * lane;
#pragma unroll
      for (int k = 0; k < 9; ++k) w[k] = Rl[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) w[9 + k] = rl[k];
    }
  }
  wave_sync();
  if (ap.accel != nullptr && (pp >= 1 || ap.accel_base)) {
    const double* __restrict__ ag = ap.accel + ((int64_t)bt * (3 * N + 1) + (pp >= 1 ? pp - 1 : 3 * N)) * N;
    if (live) s_a[lane] = ag[lane];
  } else {
    rbd::aba_tree_coop<NJ, 1, AW>(m, s_q, s_v, s_tau, s_a, s_R1, 0, lane, true);   // ends with a barrier
  }
  wave_sync();
  const double ai = live ? s_a[lane] : 0.0;
  CLK(1);
  // world-frame kinematics.  Every joint walks its own path to the root -- all lanes at once, three barriers in all -- instead of
  // the tree being swept level by level (one barrier per level, a handful of lanes busy): placement as the product of the joint
  // placements along the path, velocity and acceleration as path sums of per-joint terms.
  double oR[9], op[3], J[6], ov[6], oa[6];
  {
#pragma unroll
    for (int k = 0; k < 9; ++k) oR[k] = Rl[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) op[k] = rl[k];
    for (int a = live ? par_i : -1; a >= 0; a = s_par[a]) {   // oX_i = oX_a aX_i: rotation R_a R, translation r_a + R_a p
      const double* wa = s_W + 30 * a;
      double Ra[9], t9[9], t3[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) Ra[k] = wa[k];
      rbd::mm3(Ra, oR, t9);
      rbd::mv3(Ra, op, t3);
#pragma unroll
      for (int k = 0; k < 9; ++k) oR[k] = t9[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) op[k] = wa[9 + k] + t3[k];
    }
    double aw[3];                                         // rbdd::world_axis
    rbd::mv3(oR, ax, aw);
    if (revolute) {
      double t3[3];
      rbd::cross3(op, aw, t3);
      J[0] = aw[0]; J[1] = aw[1]; J[2] = aw[2]; J[3] = t3[0]; J[4] = t3[1]; J[5] = t3[2];
    } else {
      J[0] = 0.0; J[1] = 0.0; J[2] = 0.0; J[3] = aw[0]; J[4] = aw[1]; J[5] = aw[2];
    }
  }
  double vJ[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) vJ[k] = J[k] * vi;
  if (live) {
    double* w = s_W + 30 * lane;
#pragma unroll
    for (int k = 0; k < 6; ++k) { w[12 + k] = J[k]; w[18 + k] = vJ[k]; }
  }
  wave_sync();
#pragma unroll
  for (int k = 0; k < 6; ++k) ov[k] = vJ[k];
  for (int a = live ? par_i : -1; a >= 0; a = s_par[a]) {     // ov_i = sum over the path of J_a v_a
    const double* wa = s_W + 30 * a + 18;
#pragma unroll
    for (int k = 0; k < 6; ++k) ov[k] += wa[k];
  }
  {
    double t6[6];
    rbd::crm(ov, vJ, t6);
#pragma unroll
    for (int k = 0; k < 6; ++k) oa[k] = J[k] * ai + t6[k];   // the joint's own term of the acceleration sum
    if (live) {
      double* w = s_W + 30 * lane + 24;
#pragma unroll
      for (int k = 0; k < 6; ++k) w[k] = oa[k];
    }
  }
  wave_sync();
  for (int a = live ? par_i : -1; a >= 0; a = s_par[a]) {     // oa_i = -g + sum over the path of J_a qdd_a + ov_a x J_a v_a
    const double* wa = s_W + 30 * a + 24;
#pragma unroll
    for (int k = 0; k < 6; ++k) oa[k] += wa[k];
  }
#pragma unroll
  for (int k = 3; k < 6; ++k) oa[k] -= m.gravity[k - 3];
  CLK(2);
  // per body: world inertia, force, bias matrix (the ABA state in s_R1 is dead)
  if (live) {
    double I6[36], B[36], h[6], Ioa[6], vxh[6];
    rbdd::world_inertia(m.I6[lane], oR, op, I6);
    rbdd::m6v(I6, ov, h);
    rbdd::m6v(I6, oa, Ioa);
    rbd::crf(ov, h, vxh);
    rbdd::bias_matrix(I6, ov, h, B);
    double* o = s_R1 + 78 * lane;
#pragma unroll
    for (int k = 0; k < 36; ++k) { o[k] = I6[k]; o[36 + k] = B[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) o[72 + k] = Ioa[k] + vxh[k];
  }
  wave_sync();
  CLK(3);
  // composite sums leaves -> root: lane l < 39 owns entries 2l, 2l+1 of every record (a lane only ever touches its own entries,
  // and the LDS executes a wave's accesses in order); children precede parents in this descending sweep
  if (lane < 39) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (int i = N - 1; i >= 1; --i) {
      const int par = __builtin_amdgcn_readlane(par_i, i);
      if (par < 0) continue;
      d2* dst = reinterpret_cast<d2*>(s_R1 + 78 * par) + lane;
      const d2 c = reinterpret_cast<const d2*>(s_R1 + 78 * i)[lane];
      *dst = *dst + c;
    }
  }
  wave_sync();
  CLK(4);
  double Pr[36];                                          // y | z | u | g | Fq | Fv of the lane's joint; the others read its u | g from s_W
  if (live) {
    double* w = s_W + 30 * lane;
    const double* c = s_R1 + 78 * lane;
    double Ic[36], Bc[36], ofc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) ofc[k] = c[72 + k];
#pragma unroll
    for (int k = 0; k < 36; ++k) { Ic[k] = c[k]; Bc[k] = c[36 + k]; }
    double t1[6], t2[6], t3[6];
    rbdd::m6v(Ic, J, Pr);                                 // y
    rbdd::m6tv(Bc, J, Pr + 6);                            // z
    rbd::crm(J, ov, Pr + 12);                             // u
    rbd::crm(Pr + 12, ov, t1);
    rbd::crm(J, oa, t2);
#pragma unroll
    for (int k = 0; k < 6; ++k) Pr[18 + k] = t1[k] - t2[k];   // g
#pragma unroll
    for (int k = 0; k < 12; ++k) w[18 + k] = Pr[12 + k];
    rbd::crf(J, ofc, t1);
    rbdd::m6v(Bc, Pr + 12, t2);
    rbdd::m6v(Ic, Pr + 18, t3);
#pragma unroll
    for (int k = 0; k < 6; ++k) Pr[24 + k] = t1[k] - t2[k] + t3[k];
    rbdd::m6v(Bc, J, t1);
    rbdd::m6v(Ic, Pr + 12, t2);
#pragma unroll
    for (int k = 0; k < 6; ++k) Pr[30 + k] = t1[k] - 2.0 * t2[k];
  }
  wave_sync();
  CLK(5);
  // fused path: the resident jacobians the slabs are differenced against, and (a v direction) the trajectory point's M^-1 -- read
  // now, used after the factorisation: the latency of these reads hides behind the dynamics
  const int n = 2 * N;
  constexpr bool PRE = FUSED && NJ <= 40;
  constexpr int RTM = (NJ + 15) / 16, KSM = (NJ + 3) / 4, JTM = (2 * NJ + 15) / 16, FBN = (NJ * NJ + AW - 1) / AW;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool diff = pp > 0 && ap.write_f;
  const bool want_M = pp <= N;
  double* fx = p.fx + bt * (int64_t)n * n;
  double* fu = p.fu + bt * (int64_t)n * N;
  double base[PRE ? JTM : 1][RTM][4], fb[PRE ? FBN : 1];   // fb: f_u's lower block (a q direction) or M^-1 of the trajectory point (a v direction)
  const int step_i = AW % N, step_j = AW / N;             // (i, j) of entry k + 64 from (i, j) of entry k, k = i + j nv
  auto prefetch = [&]() {
#pragma unroll
    for (int jt = 0; jt < JTM; ++jt)
#pragma unroll
      for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 16 * jt + l4 + 4 * q, r = 16 * rt + l15;
          const bool ok = diff && j < W2 && r < N;
          base[jt][rt][q] = fx[ok ? (N + r) + j * n : 0];
        }
    int fi = lane % N, fj = lane / N;                     // entry lane + 64 u of the lower nv x nv block of f_u, without a division per entry
    const double* fbp = want_M ? fu : ap.M0 + bt * (int64_t)N * N;
#pragma unroll
    for (int u = 0; u < FBN; ++u) {
      const bool ok = want_M ? (diff && fj < N) : (lane + u * AW < N * N);
      fb[u] = fbp[ok ? (want_M ? N + fi + fj * n : lane + u * AW) : 0];
      fi += step_i; fj += step_j;
      if (fi >= N) { fi -= N; ++fj; }
    }
  };
  // (a v direction has nothing but these reads between the assembly and its products; a q direction has the factorisation, and is
  // better off issuing them there than holding 2 x 83 registers through the assembly)
  if constexpr (PRE) { if (!want_M) prefetch(); }
  // T (row-major nv x 2nv: [d tau/dq | d tau/dv]) in LDS over the dead composite region, M straight to the workspace
  double* s_T = lds + LY::T;
  for (int k = lane; k < N * W2; k += AW) s_T[k] = 0.0;
  wave_sync();
  double* Mo = ap.Mws + (sbt * (N + 1) + (want_M ? pp : 0)) * (int64_t)N * N;
  if constexpr (!FUSED) {
    if (want_M)
      for (int k = lane; k < N * N; k += AW) Mo[k] = 0.0;
    wave_sync();
  }
  double arow[FUSED ? NJ : 1];                            // fused path: row `lane` of the lower triangle of M (its nonzeros lie on the lane's path)
  if constexpr (FUSED) {
#pragma unroll
    for (int k = 0; k < NJ; ++k) arow[k] = 0.0;
  }
  CLK(6);
  uint64_t path_mask = 0;                                 // bit i: joint i is on the lane's path (itself included)
  if (live) {
    const int j = lane;
    for (int i = j; i >= 0; i = s_par[i]) {               // i in path(j): column j of row i, and (i != j) column i of row j
      path_mask |= 1ull << i;
      const double* wi = s_W + 30 * i;
      double Jr[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Jr[k] = wi[12 + k];
      double sq = 0, sv = 0, sm = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { sq += Jr[k] * Pr[24 + k]; sv += Jr[k] * Pr[30 + k]; sm += Jr[k] * Pr[k]; }
      s_T[i * W2 + j] = sq;
      s_T[i * W2 + N + j] = sv;
      if constexpr (!FUSED) {
        if (want_M) { Mo[i + (int64_t)j * N] = sm; Mo[j + (int64_t)i * N] = sm; }
      }
      if (i != j) {                                       // a proper ancestor of j: its u | g
        double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double pa = wi[18 + k]; s1 += Pr[6 + k] * pa; s2 += Pr[k] * wi[24 + k]; s3 += Pr[6 + k] * Jr[k]; s4 += Pr[k] * pa; }
        s_T[j * W2 + i] = -s1 + s2;
        s_T[j * W2 + N + i] = s3 - 2.0 * s4;
      }
    }
  }
  if constexpr (FUSED) {
    // the lane's row of M, M(j, k) = J_k . y_j for k on the path of j: a register array cannot be indexed by the run-time joint
    // number the walk above meets, so the product is formed for EVERY k (J_k: one broadcast read) and kept where the path says so
    if (want_M && live) {
#pragma unroll
      for (int k = 0; k < NJ; ++k) {                      // (k >= nv: no bit in the mask)
        const double* Jk = s_W + 30 * k + 12;
        double sm = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) sm += Jk[c] * Pr[c];
        arow[k] = ((path_mask >> k) & 1) ? sm : 0.0;
      }
    }
  }
  wave_sync();
  CLK(7);
  if constexpr (!FUSED) {
    double* To = ap.Tws + (sbt * (2 * N + 1) + pp) * (int64_t)N * W2;
    for (int k = lane; k < N * W2; k += AW) To[k] = s_T[k];
    CLK(8);
  } else {
    // ---- M^-1 of this evaluation's configuration into LDS (AnaLds::X) ----
    double* s_Mi = lds + LY::X;                           // M^-1, column r at r * nv
    if constexpr (PRE) { if (want_M) prefetch(); }
    if (want_M) {
      const int r = lane;
      // M padded to NJ x NJ by an identity block: the arithmetic on the leading nv x nv block is unchanged (the padding only ever
      // contributes exact zeros) and the factorisation is free of run-time guards
      if (r >= N && r < NJ) {
#pragma unroll
        for (int k = 0; k < NJ; ++k) arow[k] = (k == r) ? 1.0 : 0.0;
      }
      double x[NJ];
      wave_spd_inverse<INV_LDS, NJ>(arow, x, r, s_Mi);     // (the image of L sits where M^-1 goes afterwards)
      double* Mg = ap.Fws ? ap.Mws + (sbt * (N + 1) + pp) * (int64_t)N * N : nullptr;   // the constraint chain reads M^-1(q') (ana_eq_kernel)
      double* M0g = (ap.stage == 0 && ap.M0) ? ap.M0 + bt * (int64_t)N * N : nullptr;
      if (live) {
#pragma unroll
        for (int i = 0; i < NJ; ++i)
          if (i < N) {
            s_Mi[i + r * N] = x[i];                       // column r of M^-1
            if (Mg) Mg[i + (int64_t)r * N] = x[i];
            if (M0g) M0g[i + (int64_t)r * N] = x[i];
          }
      }
    } else {
      const double* M0g = ap.M0 + bt * (int64_t)N * N;    // a v direction: the trajectory point's M^-1
      double* Mg = (ap.Fws && pp == N + 1) ? ap.Mws + (sbt * (N + 1)) * (int64_t)N * N : nullptr;   // ... which ana_eq_kernel expects in slot 0
      if constexpr (PRE) {
#pragma unroll
        for (int u = 0; u < FBN; ++u) {
          const int k = lane + u * AW;
          if (k < N * N) { s_Mi[k] = fb[u]; if (Mg) Mg[k] = fb[u]; }
        }
      } else {
        for (int k = lane; k < N * N; k += AW) { const double vq = M0g[k]; s_Mi[k] = vq; if (Mg) Mg[k] = vq; }
      }
    }
    wave_sync();
    CLK(8);
    if (ap.stage == 0 && ap.m0_only) return;
    // ---- R = -M^-1 T on the matrix cores, then the outputs (the split path's ana_out_kernel, operands in LDS) ----
    typedef double f64x4_ __attribute__((ext_vector_type(4)));
    double dt = m.dt;
    const double eps = sqrt(DBL_EPSILON);
    // every global read this wave still has in flight lands here, before the first store: the stores below then never sit
    // behind a wait (the counter that guards a pending read also counts the stores issued since)
    value_fence(dt);
    if constexpr (PRE) {
#pragma unroll
      for (int jt = 0; jt < JTM; ++jt)
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q) value_fence(base[jt][rt][q]);
#pragma unroll
      for (int u = 0; u < FBN; ++u) value_fence(fb[u]);
    }
    double* slab_xx = pp > 0 ? p.fxx + (bt * n + (pp - 1)) * (int64_t)n * n : nullptr;   // f_xx(:, :, p-1): n x n
    double* slab_ux = pp > 0 ? p.fux + (bt * n + (pp - 1)) * (int64_t)n * N : nullptr;   // f_ux(:, :, p-1): n x nv
    // D'(j, r) = sum_l T(l, j) Minv(l, r): A(row = j, k = l) = T(l, j), B(k = l, col = r) = Minv(l, r); tiles 16 x 16, k by 4.
    // Result register q of a lane: D'(row = 16 jt + l4 + 4 q, col = 16 rt + l15)
    const int JT = (W2 + 15) / 16;
    // the destinations of a result are wave-uniform (bits: 1 the trajectory point's f_x, 2 its f_xx slab, 4 the workspace of the
    // constraint chain, 8 the config constraint's eq_xx slab formed on the spot): one specialised copy of the loop per
    // combination, so the loop body carries no uniform branches
    double* fws = ap.Fws ? ap.Fws + (sbt * (2 * N) + (pp - 1)) * (int64_t)N * n : nullptr;
    // config constraint (C_q = [I_e | 0], K time shifts): eq_x' = [C_q | dt C_q] f_x' has the one-term rows of ana_eq_kernel's `ident`
    // path, so the slab eq_xx(:, :, p-1) = (eq_x' - eq_x) / eps comes straight out of this wave's v rows of f_x'
    const int e_t = ap.eq_inline ? (int)p.ne[t] : 0;
    const bool eqi = e_t > 0 && pp > 0;
    const int64_t Eb = eqi ? (int64_t)b * p.d.Etot + p.Epre[t] : 0;
    const double dtK = m.eq_advance >= 2 ? 1.0 * dt : 0.0;             // C1(i, nv + i) of ana_eq_kernel
    const double* __restrict__ eqx0 = p.eq_x + Eb * n;
    double* __restrict__ eq_sxx = p.eq_xx + Eb * n * n + (int64_t)(pp - 1) * e_t * n;
    auto products = [&](auto kind_c) {
      constexpr int KIND = decltype(kind_c)::value;
      // B operand (M^-1) once for all rows of tiles; the A operand (T) of a row of tiles in one batch of LDS reads.  The rows are
      // software-pipelined: the RTM independent accumulation chains of row jt go to the matrix pipe (16 passes per
      // v_mfma_f64_16x16x4) while the VALU runs the epilogue of row jt - 1 -- sched_group_barrier interleaves the two streams
      double bv[RTM][KSM];
#pragma unroll
      for (int sk = 0; sk < KSM; ++sk)
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) {
          const int l = 4 * sk + l4, rb = 16 * rt + l15;
          bv[rt][sk] = s_Mi[(rb < N && l < N) ? l + rb * N : 0];
        }
#pragma unroll
      for (int sk = 0; sk < KSM; ++sk)
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) value_fence(bv[rt][sk]);
      batch_fence();
#pragma unroll
      for (int sk = 0; sk < KSM; ++sk)
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) bv[rt][sk] = (16 * rt + l15 < N && 4 * sk + l4 < N) ? bv[rt][sk] : 0.0;
      double av[2][KSM], oxv[2][(KIND & 8) ? RTM : 1][4];
      f64x4_ acc[2][RTM];
      auto load_row = [&](int jt, double (&a_)[KSM], double (&ox_)[(KIND & 8) ? RTM : 1][4]) {
        const int ja = 16 * jt + l15;
#pragma unroll
        for (int sk = 0; sk < KSM; ++sk) a_[sk] = s_T[(ja < W2 && 4 * sk + l4 < N) ? (4 * sk + l4) * W2 + ja : 0];
        if constexpr ((KIND & 8) != 0) {                  // eq_x at the lane's entries: requested now, used a row later
#pragma unroll
          for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int j = 16 * jt + l4 + 4 * q, r = 16 * rt + l15;
              ox_[rt][q] = eqx0[(j < W2 && r < e_t) ? r + j * e_t : 0];
            }
        }
#pragma unroll
        for (int sk = 0; sk < KSM; ++sk) value_fence(a_[sk]);
        batch_fence();
#pragma unroll
        for (int sk = 0; sk < KSM; ++sk) a_[sk] = (ja < W2 && 4 * sk + l4 < N) ? a_[sk] : 0.0;
      };
      auto mfma_row = [&](const double (&a_)[KSM], f64x4_ (&c_)[RTM]) {
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) c_[rt] = f64x4_{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sk = 0; sk < KSM; ++sk)
#pragma unroll
          for (int rt = 0; rt < RTM; ++rt)
            c_[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[sk], bv[rt][sk], c_[rt], 0, 0, 0);   // (a tile beyond nv: zero operands, no branch)
      };
      auto epilogue = [&](auto jt_c, const f64x4_ (&c_)[RTM], const double (&ox_)[(KIND & 8) ? RTM : 1][4]) {
        constexpr int jt = decltype(jt_c)::value;
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = 16 * jt + l4 + 4 * q;           // column of the jacobian block (0 .. 2nv-1: q then v directions)
            const int r = 16 * rt + l15;                  // row (joint)
            double val = (-c_[rt][q]) * dt;               // first_order_deriv, problem.hpp:499-501
            val = (j >= N && j - N == r) ? val + 1.0 : val;
            value_fence(val);                             // the jacobian entry as the reference rounds it, before it is differenced (no fused multiply-subtract)
            double bq;
            if constexpr (PRE) bq = base[jt][rt][q];
            else bq = (KIND & 2) ? fx[(j < W2 && r < N) ? (N + r) + j * n : 0] : 0.0;
            if (j < W2 && r < N) {
              const int off = (N + r) + j * n;
              if constexpr ((KIND & 1) != 0) fx[off] = val;
              if constexpr ((KIND & 2) != 0) slab_xx[off] = (val - bq) / eps;            // problem.hpp:128-137
              if constexpr ((KIND & 4) != 0) fws[r + j * N] = val;
              if constexpr ((KIND & 8) != 0) {
                if (r < e_t) {
                  const double first = j < N ? ((r == j) ? 1.0 : 0.0) * 1.0 : ((r == j - N) ? 1.0 : 0.0) * (1.0 * dt);
                  const double eqv = fma(dtK, val, first);               // ana_eq_kernel: sacc = first; sacc += C1(i, nv + i) Fv(i, j)
                  eq_sxx[r + j * e_t] = (eqv - ox_[rt][q]) / eps;        // problem.hpp:128-134
                }
              }
            }
          }
        }
      };
      load_row(0, av[0], oxv[0]);
      auto stage = [&](auto jt_c) {
        constexpr int jt = decltype(jt_c)::value;         // row jt to the matrix pipe, row jt - 1 through the epilogue, row jt + 1 requested
        if constexpr (jt < JTM) mfma_row(av[jt & 1], acc[jt & 1]);                 // (a row beyond 2 nv: zero operands, results unused)
        if constexpr (jt >= 1) { if (jt - 1 < JT) epilogue(std::integral_constant<int, jt - 1>{}, acc[(jt - 1) & 1], oxv[(jt - 1) & 1]); }
        // one matrix instruction, then the vector instructions that fit under its 16 passes
#pragma unroll
        for (int g = 0; g < RTM * KSM; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
        if constexpr (jt + 1 < JTM) load_row(jt + 1, av[(jt + 1) & 1], oxv[(jt + 1) & 1]);
      };
      static_for(stage, std::make_integer_sequence<int, JTM + 1>{});
    };
    if (pp == 0) products(std::integral_constant<int, 1>{});
    else if (eqi && ap.write_f) products(std::integral_constant<int, 2 | 8>{});
    else if (eqi) products(std::integral_constant<int, 8>{});
    else if (ap.write_f && !fws) products(std::integral_constant<int, 2>{});
    else if (ap.write_f) products(std::integral_constant<int, 2 | 4>{});
    else if (fws) products(std::integral_constant<int, 4>{});
    CLK(10);
    if (pp == 0 || ap.write_f) {
      // the rows of q+ = q + dt v are constants (problem.hpp:487-490), so their differences are exact zeros, and f_u = [0; dt M^-1]
      // (problem.hpp:493,502) has its upper nv rows zero: with an even nv the zeros of a slab go out 16 bytes per lane
      typedef double d2 __attribute__((ext_vector_type(2)));
      const bool wide = pp > 0 && (N & 1) == 0 && ((reinterpret_cast<uintptr_t>(slab_xx) | reinterpret_cast<uintptr_t>(slab_ux)) & 15) == 0;
      if (wide) {
        const int H = N / 2, si = AW % H, sj = AW / H;
        const d2 z2 = {0.0, 0.0};
        for (int i = lane % H, j = lane / H; j < n; ) {
          *reinterpret_cast<d2*>(slab_xx + 2 * i + j * n) = z2;
          i += si; j += sj;
          if (i >= H) { i -= H; ++j; }
        }
        for (int i = lane % H, j = lane / H; j < N; ) {
          *reinterpret_cast<d2*>(slab_ux + 2 * i + j * n) = z2;
          i += si; j += sj;
          if (i >= H) { i -= H; ++j; }
        }
      } else {
        for (int i = lane % N, j = lane / N; j < n; ) {
          const int off = i + j * n;
          if (pp == 0) fx[off] = (j == i) ? 1.0 : ((j == N + i) ? 1.0 * dt : 0.0);
          else slab_xx[off] = 0.0;
          i += step_i; j += step_j;
          if (i >= N) { i -= N; ++j; }
        }
      }
      CLK(11);
      // the lower nv x nv block of f_u resp. of its slab; entry idx = i + j nv
      {
        int i = lane % N, j = lane / N, idx = lane;
#pragma unroll
        for (int u = 0; u < FBN; ++u) {
          if (j < N) {
            const int off = i + j * n;
            if (pp == 0) { fu[off] = 0.0; fu[N + off] = s_Mi[idx] * dt; }
            else {
              if (!wide) slab_ux[off] = 0.0;
              if (pp <= N) {
                double fbv;
                if constexpr (PRE) fbv = fb[u]; else fbv = fu[N + off];
                double fuv = s_Mi[idx] * dt;
                value_fence(fuv);
                slab_ux[N + off] = (fuv - fbv) / eps;                       // problem.hpp:138-140
              } else slab_ux[N + off] = 0.0;              // a v direction: the same M^-1, fu_ == fu
            }
          }
          i += step_i; j += step_j; idx += AW;
          if (i >= N) { i -= N; ++j; }
        }
      }
    }
    if (eqi) {
      // eq_ux(:, :, p-1) = (eq_u' - eq_u) / eps with eq_u' = dt C_q f_u'(v rows) = dtK (dt M^-1(q')): entry idx = i + j nv of M^-1
      const double* __restrict__ equ0 = p.eq_u + Eb * N;
      double* __restrict__ eq_sux = p.eq_ux + Eb * N * n + (int64_t)(pp - 1) * e_t * N;
      double ouv[FBN];
      {
        int i = lane % N, j = lane / N;
#pragma unroll
        for (int u = 0; u < FBN; ++u) {
          ouv[u] = equ0[(j < N && i < e_t) ? i + j * e_t : 0];
          i += step_i; j += step_j;
          if (i >= N) { i -= N; ++j; }
        }
      }
      int i = lane % N, j = lane / N, idx = lane;
#pragma unroll
      for (int u = 0; u < FBN; ++u) {
        if (j < N && i < e_t) {
          double fuv = s_Mi[idx] * dt;
          value_fence(fuv);
          eq_sux[i + j * e_t] = (fma(dtK, fuv, 0.0) - ouv[u]) / eps;   // problem.hpp:135-137
        }
        i += step_i; j += step_j; idx += AW;
        if (i >= N) { i -= N; ++j; }
      }
    }
    CLK(9);
  }
#ifdef DEV_ANA_CLOCKS
  if (lane == 0 && ap.stage == 1 && (e == 5 * P + 3 || e == 17 * P + 50 || e == 100 * P + 1))
    printf("ana_eval pp=%d: setup %llu aba/load %llu levels %llu body1 %llu composite %llu body2 %llu zero %llu assemble %llu store/minv %llu out %llu (mfma %llu zero %llu fu %llu) total %llu\n", pp,
           clk[1] - clk[0], 0ull, clk[2] - clk[1], clk[3] - clk[2], clk[4] - clk[3], clk[5] - clk[4], clk[6] - clk[5], clk[7] - clk[6], clk[8] - clk[7], clk[9] - clk[8], clk[10] - clk[8], clk[11] - clk[10], clk[9] - clk[11], (FUSED ? clk[9] : clk[8]) - clk[0]);
#endif
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
        value_fence(val);
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
    else if (pp <= N) {
      double fuv = i < N ? 0.0 : Mi[(i - N) + (int64_t)j * N] * dt;
      value_fence(fuv);
      slab_ux[off] = i < N ? 0.0 : (fuv - fu[off]) / eps;               // problem.hpp:138-140
    }
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
  double* s_st = lds;                                   // ABA state (absent when the launch runs no forward dynamics: AnaParams::eq_no_aba)
  double* s_q = s_st + (ap.eq_no_aba ? 0 : rbd::ABA_LDS_SLOTS * NJ);
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
  // the look-ahead states x_1 .. x_K (dynamics_t::eval_to, problem.hpp:441-461).  Both base constraints read q alone, so the last
  // step needs q_K = q_{K-1} + dt v_{K-1} and no forward dynamics; the identity jacobian of the config constraint does not depend
  // on x_K at all (a perturbed direction then needs no look-ahead state); and the acceleration at the perturbed point itself is
  // the one the static first-order kernels formed for the dynamics' own pass (AnaParams::accel)
  const bool need_state = dir == 0 || m.eq_kind != DDP_HIP_EQ_CONFIG;
  for (int k = 0; k < K && need_state; ++k) {
    const bool last = k == K - 1;
    if (!last) {
      if (k == 0 && dir >= 1 && ap.eq_no_aba) {
        const double* __restrict__ ag = ap.accel + ((int64_t)bt * (3 * N + 1) + (dir - 1)) * N;
        for (int i = lane; i < N; i += AW) s_a[i] = ag[i];
        __syncthreads();
      } else {
        rbd::aba_tree_coop<NJ, 1, AW>(m, s_q, s_v, s_tau, s_a, s_st, 0, lane, true);   // ends with a barrier
      }
    }
    for (int i = lane; i < N; i += AW) {
      const double vo = dt * s_v[i];
      s_q[i] = s_q[i] + vo;
      if (!last) s_v[i] = s_v[i] + s_a[i] * dt;
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
  const double* __restrict__ ox = p.eq_x + Eb * n;
  const double* __restrict__ ou = p.eq_u + Eb * N;
  // entry idx = i + j e of an e-row block: (i, j) advance by (64 mod e, 64 div e) per pass, no division per entry
  const int si = AW % e, sj = AW / e;
  // config constraint: C_q = [I_e | 0], so row i of a product has ONE non-zero term (r = i); every other term of the reference's
  // sum is 0 * finite, which leaves the running sum as it is
  const bool ident = m.eq_kind == DDP_HIP_EQ_CONFIG;
  if (dir <= 2 * N) {
    const int idx3 = dir - 1;                             // the right index of the slab
    const double* __restrict__ Fv = ap.Fws + (sbt * (2 * N) + idx3) * (int64_t)N * n;                   // v rows of f_x at the perturbed point
    const double* __restrict__ Mi = ap.Mws + (sbt * (N + 1) + (dir <= N ? dir : 0)) * (int64_t)N * N;  // M^-1 of its configuration
    double* __restrict__ sxx = p.eq_xx + Eb * n * n + (int64_t)idx3 * e * n;
    double* __restrict__ sux = p.eq_ux + Eb * N * n + (int64_t)idx3 * e * N;
    {
      int i = lane % e, j = lane / e;
#pragma unroll 4
      for (int idx = lane; idx < e * n; idx += AW) {
        // the q rows of f_x are the constants [I | dt I] (problem.hpp:487-490): one non-zero term, the others exact zeros
        double sacc = j < N ? C1(i, j) * 1.0 : C1(i, j - N) * (1.0 * dt);
        if (ident) sacc = fma(C1(i, N + i), Fv[i + j * N], sacc);
        else for (int r = 0; r < N; ++r) sacc += C1(i, N + r) * Fv[r + j * N];
        sxx[idx] = (sacc - ox[idx]) / eps;                // problem.hpp:128-134
        i += si; j += sj;
        if (i >= e) { i -= e; ++j; }
      }
    }
    {
      int i = lane % e, j = lane / e;
#pragma unroll 4
      for (int idx = lane; idx < e * N; idx += AW) {
        double sacc = 0.0;                                // the q rows of f_u are zero (problem.hpp:493)
        if (ident) { double fuv = Mi[i + j * N] * dt; value_fence(fuv); sacc = fma(C1(i, N + i), fuv, sacc); }
        else for (int r = 0; r < N; ++r) sacc += C1(i, N + r) * (Mi[r + j * N] * dt);
        sux[idx] = (sacc - ou[idx]) / eps;                // problem.hpp:135-137
        i += si; j += sj;
        if (i >= e) { i -= e; ++j; }
      }
    }
  } else {
    const int idx3 = dir - 1 - 2 * N;
    double* __restrict__ suu = p.eq_uu + Eb * N * N + (int64_t)idx3 * e * N;
    const double* __restrict__ fur = fu;
    int i = lane % e, j = lane / e;
#pragma unroll 4
    for (int idx = lane; idx < e * N; idx += AW) {
      double sacc = 0.0;
      if (ident) { sacc += C1(i, i) * fur[i + j * n]; sacc += C1(i, N + i) * fur[N + i + j * n]; }
      else for (int l = 0; l < n; ++l) sacc += C1(i, l) * fur[l + j * n];   // f_u(x, u + eps e) == f_u(x, u), bit for bit
      suu[idx] = (sacc - ou[idx]) / eps;                  // problem.hpp:141-145
      i += si; j += sj;
      if (i >= e) { i -= e; ++j; }
    }
  }
}

template <int NJ>
size_t eq_lds_bytes(const Dims& d, bool no_aba = false) { return sizeof(double) * (size_t)((no_aba ? 0 : rbd::ABA_LDS_SLOTS * NJ) + 4 * NJ + d.emax * NJ); }

// flags: ANA_F the dynamics' own outputs (stage 0: f_x, f_u; stage 1: f_xx, f_ux, f_uu), ANA_EQ the constraint chain's
// (stage 0: eq_val, eq_x, eq_u from the resident f_x, f_u; stage 1: eq_xx, eq_ux, eq_uu)
template <int NJ>
int launch_t(ddp_hip_ctx* ctx, const LinParams& p, int stage, int flags) {
  const Dims& d = ctx->d;
  const int64_t BT = d.batch * d.T;
  const int N = (int)d.nv;
  const size_t lds = sizeof(double) * (size_t)AnaLds<NJ>::TOTAL;
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
  if (stage == 1 && do_f && !ctx->fuu_zero) {
    // f_uu is exactly zero (see the header of this file); it stays so until someone else writes the sequence (ctx.hip clears the flag)
    HIP_TRY(hipMemsetAsync(p.fuu, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_FUU].size * d.batch), ctx->stream));
    ctx->fuu_zero = true;
  }
  const bool accel_now = stage == 0 && (flags & ANA_ACCEL) != 0 && do_f;
  if ((stage == 1 || accel_now) && ctx->ana_A && ctx->lin_static && p.qcache) {
    // the forward dynamics of the 2 nv perturbed points: the static first-order kernels evaluate exactly these points
    // (x + sqrt(eps_mach) e_k) chain-wise from the base point's cache, ~25x cheaper than one cooperative ABA per point.  When the
    // mode-1 pass follows in the same linearisation call they are formed ahead of stage 0, which then takes the trajectory point's
    // own acceleration from the same kernels (the idle lanes of the v-level wave evaluate it) instead of running an ABA per point
    if (!ctx->ana_A_fresh) {
      LinParams pa = p;
      pa.accel_out = ctx->ana_A;
      const bool with_u = stage == 1 ? do_eq : ((flags & ANA_EQ_NEXT) != 0 && d.Etot > 0);   // the constraint chain also differences along the u directions
      const int rc_ = lin_static_launch(ctx, pa, with_u ? 7 : 6);
      if (rc_ != DDP_HIP_OK) return rc_;
      ctx->ana_A_fresh = accel_now;                       // (lin.hip drops the mark when the linearisation call returns)
    }
    ap.accel = ctx->ana_A;
    ap.accel_base = accel_now ? 1 : 0;
    ap.eq_no_aba = (stage == 1 && do_eq && ctx->model_h.eq_advance <= 2) ? 1 : 0;   // K <= 2: the one acceleration a direction needs is in ana_A
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
  if (!ctx->ana_split) {
    // fused path: one wave takes an evaluation from the state to its output columns; no T / M workspace
    ap.M0 = ctx->ana_M0;
    if (stage == 0) {
      ap.bt0 = 0; ap.nbt = (int32_t)BT;
      hipLaunchKernelGGL((ana_eval_kernel<NJ, true>), dim3((unsigned)BT), dim3(AW), lds, ctx->stream, ap);
      ctx->ana_M0_fresh = ctx->ana_M0 != nullptr;         // (lin.hip drops the mark when the linearisation call returns)
      if (do_eq) hipLaunchKernelGGL((ana_eq_kernel<NJ>), dim3((unsigned)BT), dim3(AW), eq_lds_bytes<NJ>(d), ctx->stream, ap);
      HIP_TRY(hipGetLastError());
      return DDP_HIP_OK;
    }
    if (!ctx->ana_M0) return DDP_HIP_E_UNSUPPORTED;
    if (!ctx->ana_M0_fresh) {                             // M^-1 at the trajectory points, for the v directions (else: stage 0 of this very linearisation left it)
      AnaParams a0 = ap;
      a0.stage = 0; a0.m0_only = 1; a0.bt0 = 0; a0.nbt = (int32_t)BT; a0.Fws = nullptr; a0.accel = nullptr;
      hipLaunchKernelGGL((ana_eval_kernel<NJ, true>), dim3((unsigned)BT), dim3(AW), lds, ctx->stream, a0);
    }
    // config constraint: its x-direction tensors come out of the evaluation waves themselves and eq_uu is exactly zero (f_u does not
    // depend on u and C is constant: ana_eq_kernel forms the same sum twice and differences it) -- no workspace, no slices
    const bool eq_inline = do_eq && ctx->model_h.eq_kind == DDP_HIP_EQ_CONFIG && getenv("DDP_HIP_ANA_EQ_KERNEL") == nullptr;
    if (eq_inline) {
      ap.Fws = nullptr;
      ap.eq_inline = 1;
      HIP_TRY(hipMemsetAsync(p.eq_uu, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_UU].size * d.batch), ctx->stream));
    }
    const bool eq_kernel = do_eq && !eq_inline;
    const int64_t step = eq_kernel ? ctx->ana_nbt : BT;   // the constraint chain reads per-slice workspaces (Fws, Mws)
    for (int64_t bt0 = 0; bt0 < BT; bt0 += step) {
      const int64_t nb = BT - bt0 < step ? BT - bt0 : step;
      ap.bt0 = bt0;
      ap.nbt = (int32_t)nb;
      hipLaunchKernelGGL((ana_eval_kernel<NJ, true>), dim3((unsigned)(nb * 2 * N)), dim3(AW), lds, ctx->stream, ap);
      if (eq_kernel) hipLaunchKernelGGL((ana_eq_kernel<NJ>), dim3((unsigned)(nb * 3 * N)), dim3(AW), eq_lds_bytes<NJ>(d, ap.eq_no_aba != 0), ctx->stream, ap);
    }
    HIP_TRY(hipGetLastError());
    return DDP_HIP_OK;
  }
  for (int64_t bt0 = 0; bt0 < BT; bt0 += ctx->ana_nbt) {
    const int64_t nb = BT - bt0 < ctx->ana_nbt ? BT - bt0 : ctx->ana_nbt;
    ap.bt0 = bt0;
    ap.nbt = (int32_t)nb;
    hipLaunchKernelGGL((ana_eval_kernel<NJ, false>), dim3((unsigned)(nb * P)), dim3(AW), lds, ctx->stream, ap);
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
  ctx->ana_split = getenv("DDP_HIP_ANA_SPLIT") != nullptr;    // development: the three-kernel form with its HBM workspaces
  const bool m1 = ctx->model_h.fd_mode == 1 && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS);
  if (ctx->ana_split) HIP_TRY(hipMalloc(&ctx->ana_T, sizeof(double) * (size_t)(ctx->ana_nbt * (2 * N + 1) * N * 2 * N)));
  if (ctx->ana_split || (m1 && d.Etot > 0)) HIP_TRY(hipMalloc(&ctx->ana_M, sizeof(double) * (size_t)(ctx->ana_nbt * (N + 1) * N * N)));
  if (!ctx->ana_split && m1) HIP_TRY(hipMalloc(&ctx->ana_M0, sizeof(double) * (size_t)(BT * N * N)));
  if (ctx->lin_static && ctx->lin_ws && ctx->model_h.fd_mode == 1 && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS))
    HIP_TRY(hipMalloc(&ctx->ana_A, sizeof(double) * (size_t)(BT * (3 * N + 1) * N)));
  if (d.Etot > 0) {
    // the constraint chain on analytic jacobians (ana_eq_kernel): K <= 2 look-ahead steps, see the kernel's header
    if (ctx->model_h.eq_advance < 1 || ctx->model_h.eq_advance > 2) return DDP_HIP_E_UNSUPPORTED;
    if (ctx->model_h.fd_mode == 1 && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS))
      HIP_TRY(hipMalloc(&ctx->ana_F, sizeof(double) * (size_t)(ctx->ana_nbt * 2 * N * N * 2 * N)));
    const size_t l38 = eq_lds_bytes<38>(d), l64 = eq_lds_bytes<64>(d);
    if (l38 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ana_eq_kernel<38>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l38));
    if (d.nv > 38 && l64 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ana_eq_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l64));
  }
  if (d.nv > 38) {
    const size_t lds = sizeof(double) * (size_t)AnaLds<64>::TOTAL;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ana_eval_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ana_eval_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  return DDP_HIP_OK;
}

void lin_analytic_teardown(ddp_hip_ctx* ctx) {
  if (ctx->ana_T) (void)hipFree(ctx->ana_T);
  if (ctx->ana_M) (void)hipFree(ctx->ana_M);
  if (ctx->ana_F) (void)hipFree(ctx->ana_F);
  if (ctx->ana_A) (void)hipFree(ctx->ana_A);
  if (ctx->ana_M0) (void)hipFree(ctx->ana_M0);
}

int lin_analytic_launch(ddp_hip_ctx* ctx, const LinParams& p, int stage, int flags) {
  if (ctx->ana_nbt <= 0) return DDP_HIP_E_UNSUPPORTED;   // lin_analytic_setup did not take this model
  if (stage == 1 && !p.has_tensors) return DDP_HIP_OK;
  if (ctx->d.nv <= 38) return launch_t<38>(ctx, p, stage, flags);
  return launch_t<64>(ctx, p, stage, flags);
}
