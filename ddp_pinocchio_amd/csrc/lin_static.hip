// lin_static.hip -- second-order finite differences (finite_diff_hessian_compute mode 2, problem.hpp:152-298) for
// trees whose TOPOLOGY is known at compile time.
//
// The generic kernels of lin.hip keep the per-joint state of an evaluation in private arrays indexed by run-time
// joint / slot numbers; those arrays live in scratch (HBM-backed) and their traffic is what bounds the stencil
// (profiles/: 39 KB written per evaluation at the Talos size).  Here the parent table is a template parameter: every
// loop over the joints is expanded at compile time, every index is a constant and the running state of an
// evaluation lives in registers.  One wave = 64 stencil points that share their configuration q (and, at the
// torque level, their velocity): what the articulated-body algorithm derives from q (and v) alone is read from the
// q- / v-caches through the scalar path (wave-uniform addresses), each lane only carries what its own perturbation
// changes.  The arithmetic is the operation sequence of rbd::aba_u_cached / rbd::aba_vu_cached.
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "internal.h"
#include "lin_common.h"
#include "rbd.h"

#ifndef DEV_NO_EMIT
#define DEV_NO_EMIT 0
#endif
namespace {

// ---- compiled-in topologies -------------------------------------------------------------------------------------
// Talos-like humanoid (models.cpp:build_tree38): floating base as 3 prismatic + 3 revolute joints, two 6-joint legs,
// a 2-joint torso, two 8-joint arms, a 2-joint head
struct TopoTalos38 {
  static constexpr int N = 38;
  static constexpr int parent[N] = {-1, 0, 1, 2, 3, 4,            // base
                                    5, 6, 7, 8, 9, 10,            // left leg
                                    5, 12, 13, 14, 15, 16,        // right leg
                                    5, 18,                        // torso
                                    19, 20, 21, 22, 23, 24, 25, 26,   // left arm
                                    19, 28, 29, 30, 31, 32, 33, 34,   // right arm
                                    19, 36};                      // head
  static constexpr int prismatic[N] = {1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                       0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

// UR5-like serial arm (models.cpp:build_chain6; test/pinocchio_ddp.cpp shape): one chain of six revolute joints
struct TopoChain6 {
  static constexpr int N = 6;
  static constexpr int parent[N] = {-1, 0, 1, 2, 3, 4};
  static constexpr int prismatic[N] = {0, 0, 0, 0, 0, 0};
};

}  // namespace
#include "topo_extra.h"     // generated topologies (tools/gen_topology.py): struct Topo<Name> ..., DDP_TOPO_EXTRA(X)
namespace {

template <class T> constexpr bool has_child(int k) {
  for (int c = k + 1; c < T::N; ++c) if (T::parent[c] == k) return true;
  return false;
}
template <class T> constexpr int largest_child(int k) {
  int r = -1;
  for (int c = k + 1; c < T::N; ++c) if (T::parent[c] == k) r = c;
  return r;
}
// the largest-index child of its parent contributes first in the leaf -> root pass
template <class T> constexpr bool first_contrib(int k) { return T::parent[k] >= 0 && largest_child<T>(T::parent[k]) == k; }

template <class T>
bool topo_matches(const DevModel& m) {
  if (m.kind != DDP_HIP_MODEL_TREE || m.nv != T::N) return false;
  for (int i = 0; i < T::N; ++i)
    if (m.parent[i] != T::parent[i] || (m.jtype[i] == DDP_HIP_JOINT_PRISMATIC) != (T::prismatic[i] != 0)) return false;
  return true;
}

// ---- per-lane state ----------------------------------------------------------------------------------------------
template <class T>
struct TauState {
  double acc[T::N][6];   // leaf -> root: bias-force accumulators of the joints that have children; root -> leaf: accelerations
  double uu[T::N];       // u_i = tau_i - S_i^T pA_i, then the joint acceleration
};

// leaf -> root step of joint K for a new tau (rbd::aba_u_cached, first loop)
// qc: the (E | r | U | 1/D) part of the q-cache, QS doubles per joint (the cache itself, or a copy of it in LDS)
template <class T, int K, int QS>
__device__ __forceinline__ void tau_up(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ vc,
                                       double tauK, TauState<T>& s) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* E = qc + K * QS;
  const double* r = E + 9;
  const double* U = E + 12;
  const double dinv = E[18];
  const double* pA0 = vc + K * rbd::VC_STRIDE + 6;
  const double* Iac = vc + K * rbd::VC_STRIDE + 12;
  const double* a = m.axis[K];
  double pAi[6];
  if constexpr (has_child<T>(K)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) pAi[k] = s.acc[K][k];
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) pAi[k] = pA0[k];
  }
  double sp = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) sp += a[k] * pAi[o + k];
  const double ui = tauK - sp;
  s.uu[K] = ui;
  constexpr int par = T::parent[K];
  if constexpr (par >= 0) {
    double pa[6], fp[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
    rbd::xform_force_T(E, r, pa, fp);
    if constexpr (first_contrib<T>(K)) {
      const double* pp0 = vc + par * rbd::VC_STRIDE + 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) s.acc[par][k] = pp0[k] + fp[k];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) s.acc[par][k] += fp[k];
    }
  }
  if constexpr (QS == rbd::QC_STRIDE || (K % 2) == 0) __builtin_amdgcn_sched_barrier(0);   // keep the operand loads of the next joint out of this one: SGPRs would spill
}

// root -> leaf step of joint K (rbd::aba_u_cached, second loop); leaves the joint acceleration in uu[K]
template <class T, int K, int QS>
__device__ __forceinline__ void tau_down(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ vc,
                                         TauState<T>& s) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* E = qc + K * QS;
  const double* r = E + 9;
  const double* U = E + 12;
  const double dinv = E[18];
  const double* cb = vc + K * rbd::VC_STRIDE;
  double ap[6];
  constexpr int par = T::parent[K];
  if constexpr (par >= 0) rbd::xform_motion(E, r, s.acc[par], ap);
  else {
    const double a0[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
    rbd::xform_motion(E, r, a0, ap);
  }
  double sum = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; sum += U[k] * ap[k]; }
  const double qd = (s.uu[K] - sum) * dinv;
  s.uu[K] = qd;
  if constexpr (has_child<T>(K)) {
    const double* a = m.axis[K];
#pragma unroll
    for (int k = 0; k < 6; ++k) s.acc[K][k] = ap[k];
    s.acc[K][o] += a[0] * qd; s.acc[K][o + 1] += a[1] * qd; s.acc[K][o + 2] += a[2] * qd;
  }
  if constexpr (QS == rbd::QC_STRIDE || (K % 2) == 0) __builtin_amdgcn_sched_barrier(0);
}

template <class T, int QS, class TauFn, int... Ks>
__device__ __forceinline__ void tau_up_all(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ vc, TauFn tau,
                                           TauState<T>& s, std::integer_sequence<int, Ks...>) {
  (tau_up<T, T::N - 1 - Ks, QS>(m, qc, vc, tau(T::N - 1 - Ks), s), ...);
}
// the joints HI, HI - 1, ..., LO of the leaf -> root pass
template <class T, int QS, int HI, class TauFn, int... Ks>
__device__ __forceinline__ void tau_up_range(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ vc, TauFn tau,
                                             TauState<T>& s, std::integer_sequence<int, Ks...>) {
  (tau_up<T, HI - Ks, QS>(m, qc, vc, tau(HI - Ks), s), ...);
}
template <class T, int QS, int... Ks>
__device__ __forceinline__ void tau_down_all(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ vc,
                                             TauState<T>& s, std::integer_sequence<int, Ks...>) {
  (tau_down<T, Ks, QS>(m, qc, vc, s), ...);
}

// copy of the (E | r | U | 1/D) part of a q-cache block in LDS, PS doubles per joint: the scalar path fills at about
// 2 bytes per clock and CU, so operands that both tree passes need are read from LDS instead (as broadcast reads)
constexpr int PS = 20;
template <int NV>
__device__ __forceinline__ void stage_placements(double* sp, const double* __restrict__ qc, int lane) {
  for (int idx = lane; idx < NV * 19; idx += LBS) {
    const int K = idx / 19, e = idx - K * 19;
    sp[K * PS + e] = qc[K * rbd::QC_STRIDE + e];
  }
}

// Pull a wave-uniform block into the L2 ahead of the scalar loads that walk it: 64 lanes x 16 bytes per KiB, all in
// flight at once (the scalar path alone would pay the HBM latency once per 64-byte line, one line after the other).
// Returns a word that depends on every byte loaded; the caller folds it into the start of its dependency chain so that
// the evaluation begins once the block has arrived.  Plain loads on purpose: inline asm or a memory-writing intrinsic
// anywhere ahead of the operand loads makes the compiler fall back from scalar to per-lane vector loads.
template <int BYTES>
__device__ __forceinline__ unsigned int warm_block(const double* base, int lane) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const char* b = reinterpret_cast<const char*>(base);
  unsigned int sink = 0;
#pragma unroll
  for (int off = 0; off < BYTES; off += LBS * 16) {
    int o = off + lane * 16;
    o = o < BYTES - 16 ? o : BYTES - 16;
    const u4 v = *reinterpret_cast<const u4*>(b + o);
    sink ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  return sink;
}

template <class T> constexpr bool parents_at_least(int from, int lo) {
  for (int k = from; k < T::N; ++k) if (T::parent[k] < lo) return false;
  return true;
}
// where the two-phase staging of a row's operands splits the joints: the largest KH <= N / 2 such that the joints above KH
// only touch records KH .. N-1 (their own and their parents'); 19 for the Talos-like tree (its arms hang from joint 19),
// 0 (no second phase) for a tree whose upper half reaches back to the root
template <class T> constexpr int stage_split() {
  for (int kh = T::N / 2; kh > 0; --kh) if (parents_at_least<T>(kh + 1, kh)) return kh;
  return 0;
}

// Two-phase staging of a row's operands (E | r | U | 1/D from the q-cache block, the whole v-cache block): the records of the
// joints KH .. NV-1 -- where the leaf -> root pass starts -- are waited for and parked first; the loads of the joints 0 .. KH-1
// are issued at the same time, stay in flight (in registers) while the first half of the pass runs and are parked in front of
// the second half.  A row wave otherwise spends a quarter of its life waiting for its 17 KB of operands before it does anything.
template <int NV, int K0, int K1, bool WITH_V = true>
struct StageRegs {
  static constexpr int NPW = (K1 - K0) * 19, NVW = WITH_V ? (K1 - K0) * rbd::VC_STRIDE : 0;
  static constexpr int CP = (NPW + LBS - 1) / LBS, CV = (NVW + LBS - 1) / LBS;
  double rp[CP], rv[CV > 0 ? CV : 1];
  __device__ __forceinline__ void load(const double* __restrict__ qc, const double* __restrict__ vc, int lane) {
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      int idx = c * LBS + lane;
      idx = idx < NPW ? idx : NPW - 1;
      const int K = K0 + idx / 19, e = idx % 19;
      rp[c] = qc[K * rbd::QC_STRIDE + e];
    }
#pragma unroll
    for (int c = 0; c < CV; ++c) {
      int idx = c * LBS + lane;
      idx = idx < NVW ? idx : NVW - 1;
      rv[c] = vc[K0 * rbd::VC_STRIDE + idx];
    }
  }
  // (no branches: the lanes past the end store the last word again -- control flow in the middle of the evaluation would split
  // its basic block, and the optimiser then moves it away from its operand loads, see the notes on scalar-path hygiene)
  __device__ __forceinline__ void park(double* sp, double* sv, int lane) const {
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      int idx = c * LBS + lane;
      idx = idx < NPW ? idx : NPW - 1;
      const int K = K0 + idx / 19, e = idx % 19;
      sp[K * PS + e] = rp[c];
    }
#pragma unroll
    for (int c = 0; c < CV; ++c) {
      int idx = c * LBS + lane;
      idx = idx < NVW ? idx : NVW - 1;
      sv[K0 * rbd::VC_STRIDE + idx] = rv[c];
    }
  }
};

__device__ __forceinline__ void tri_index(int64_t q, int Wd, int& ii, int& jj) {   // q -> (ii, jj), ii < jj < Wd, row by row
  int a = (int)floor(((2.0 * Wd - 1.0) - sqrt((2.0 * Wd - 1.0) * (2.0 * Wd - 1.0) - 8.0 * (double)q)) * 0.5);
  if (a < 0) a = 0;
  while ((int64_t)a * (2 * Wd - a - 1) / 2 > q) --a;
  while ((int64_t)(a + 1) * (2 * Wd - a - 2) / 2 <= q) ++a;
  ii = a;
  jj = (int)(q - (int64_t)a * (2 * Wd - a - 1) / 2) + a + 1;
}

// ---- output stage of the pair kernels (problem.hpp:226-296) -----------------------------------------------------------
// NP: points per pass of the output stage (LBS: all of the wave's at once; LBS / 2: the torque-level pair kernel, whose
// accelerations are in registers until then, emits its two half-waves one after the other and needs half the LDS)
template <int NV, int NP = LBS>
struct OutStage {
  double qdd[NP * (NV + 1)];    // point-major, odd row stride: (joint k, point e) at qdd[e * (NV + 1) + k] -- conflict-free for the
                                // evaluation (lane = point) and for the output stage (16 lanes = 16 consecutive joints of one point)
  int pij[NP];                  // the point: first direction | second direction << 8 | valid << 16 (the six column addresses of a
                                // point are functions of these and of wave-uniform bases: recomputed by the lanes that emit it)
};

// Each stencil point owns one n-double column of a tensor (two for a symmetric pair) and reads four more columns (its two
// first-order columns and its two diagonal second-order columns).  Every lane leaves the accelerations of its evaluation and
// the six column addresses of its point in LDS.  Then a 16-lane group takes one point at a time: the lanes read the point's
// addresses once (LDS broadcasts) and own the rows k = kk + 16 j of its columns -- x_k, f_k sit in registers for the whole
// stage, an element costs four loads, at most one (conflict-free) LDS read and the store(s); two points per group in flight.
// (Round 1's form walked four points per 16-row chunk and re-read the addresses for every chunk: 8 LDS reads per element, the
// acceleration among them 16-way bank-conflicted; the stage was 12.5 ms of the 86 ms linearisation.)
// pass: which NP-block of the wave's lanes owns the slots this time (the caller has put their accelerations into S.qdd)
template <int NV, int NP>
__device__ __forceinline__ void offdiag_emit(const LinParams& p, OutStage<NV, NP>& S, bool valid, int i, int j, int64_t bt,
                                             const double* __restrict__ xg, double dt, int pass = 0) {
  constexpr int n = 2 * NV, mm = NV;
  // points in flight per group: two; one where the caller still holds the accelerations of the second half in registers
  constexpr int RW = 16, NJ = (n + RW - 1) / RW, GRP = LBS / RW, PPG = NP / GRP, NB = NP == LBS ? 2 : 1;
  static_assert(PPG % NB == 0, "points per group");
  const double eps = sqrt(sqrt(DBL_EPSILON));
  const double eps2 = eps * eps;
  const int lane = threadIdx.x;
  if (lane / NP == pass) S.pij[lane % NP] = i | (j << 8) | ((valid ? 1 : 0) << 16);
  double* const fxx = p.fxx + bt * n * n * n;
  double* const fux = p.fux + bt * n * mm * n;
  double* const fuu = p.fuu + bt * n * mm * mm;
  const double* const fxb = p.fx + bt * n * n;
  const double* const fub = p.fu + bt * n * mm;
  __syncthreads();
  const int kk = lane % RW, grp = lane / RW;
  const double* f0 = p.f_val + bt * n;
  double f0k[NJ], xk[NJ], xvk[NJ];
#pragma unroll
  for (int jx = 0; jx < NJ; ++jx) {
    const int k = kk + RW * jx;
    const int kc = k < n ? k : n - 1;
    f0k[jx] = f0[kc];
    xk[jx] = xg[kc];
    xvk[jx] = kc < NV ? xg[NV + kc] : 0.0;
  }
  if (DEV_NO_EMIT) return;
  for (int pt = 0; pt < PPG; pt += NB) {
    double a1[NB][NJ], a2[NB][NJ], d1[NB][NJ], d2[NB][NJ];
    double* o0[NB];
    double* o1[NB];
    int ie[NB], je[NB], ee[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = (pt + u) * GRP + grp;
      ee[u] = e;
      const int pk = S.pij[e];
      const int i1 = pk & 255, j1 = (pk >> 8) & 255;
      ie[u] = i1;
      je[u] = j1;
      const bool at_x_1 = i1 < n, at_x_2 = j1 < n;
      const int idx_1 = at_x_1 ? i1 : i1 - n, idx_2 = at_x_2 ? j1 : j1 - n;
      double* tensor;
      int L;
      if (at_x_1) { if (at_x_2) { tensor = fxx; L = n; } else { tensor = fux; L = mm; } }
      else { tensor = fuu; L = mm; }
      o0[u] = (pk >> 16) ? tensor + (int64_t)idx_2 * n + (int64_t)idx_1 * n * L : nullptr;
      o1[u] = ((pk >> 16) && at_x_1 == at_x_2 && !p.skip_qv_mirror) ? tensor + (int64_t)idx_1 * n + (int64_t)idx_2 * n * L : nullptr;
      if (o0[u]) {
        const double* q0 = at_x_1 ? fxb + (int64_t)idx_1 * n : fub + (int64_t)idx_1 * n;
        const double* q1 = at_x_2 ? fxb + (int64_t)idx_2 * n : fub + (int64_t)idx_2 * n;
        const double* q2 = at_x_1 ? fxx + (int64_t)idx_1 * n + (int64_t)idx_1 * n * n : fuu + (int64_t)idx_1 * n + (int64_t)idx_1 * n * mm;
        const double* q3 = at_x_2 ? fxx + (int64_t)idx_2 * n + (int64_t)idx_2 * n * n : fuu + (int64_t)idx_2 * n + (int64_t)idx_2 * n * mm;
#pragma unroll
        for (int jx = 0; jx < NJ; ++jx) {
          const int k = kk + RW * jx;
          const int kc = k < n ? k : n - 1;
          a1[u][jx] = q0[kc]; a2[u][jx] = q1[kc]; d1[u][jx] = q2[kc]; d2[u][jx] = q3[kc];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (!o0[u]) continue;
#pragma unroll
      for (int jx = 0; jx < NJ; ++jx) {
        const int k = kk + RW * jx;
        if (k >= n) continue;
        // configuration rows: q+ = q + dt v does not see the dynamics; row k < nv of a stencil point differs from the base point's
        // only when one of the two directions is q_k or v_k -- every other entry is an exact zero, already in memory (skip_top)
        if (p.skip_top && k < NV && !((ie[u] < n && ie[u] % NV == k) || (je[u] < n && je[u] % NV == k))) continue;
        // row k of f(x + dx, u + du) (dynamics_t::eval_to, problem.hpp:441-461)
        double xs = k == ie[u] ? xk[jx] + eps : xk[jx];
        if (k == je[u]) xs = xs + eps;
        double fv;
        if (k < NV) {
          double xv = (NV + k) == ie[u] ? xvk[jx] + eps : xvk[jx];
          if (NV + k == je[u]) xv = xv + eps;
          const double vo = dt * xv;
          fv = xs + vo;
        } else {
          fv = xs + S.qdd[ee[u] * (NV + 1) + (k - NV)] * dt;
        }
        double df = fv - f0k[jx];                         // difference_out
        df -= eps * a1[u][jx];
        df -= eps * a2[u][jx];
        df *= 2;
        const double val = 0.5 * (df / eps2 - d1[u][jx] - d2[u][jx]);
        o0[u][k] = val;
        if (o1[u]) o1[u][k] = val;
      }
    }
  }
}

// ---- output stage for a row of the stencil: first direction i shared by the wave, second direction jb + lane -----
// The NV columns such a wave owns are one contiguous block of its tensor (element (k, lane) at k + lane n), and so is the
// block of first-order columns they read.  The accelerations are left in LDS lane-major with an odd row stride.
template <int NV>
struct RowStage {
  // rows 0 .. NV-1: the row's points; row NV: the torque-level kernel's extra point (direction i alone: DIAG_ROW below);
  // row NV+1: where the idle lanes of the wave leave their values.  12.5 KB at NV = 38: twelve workgroups per CU
  double q[(NV + 2) * (NV + 1)];
};

// Lane = (column group cg = lane / 16, kk = lane % 16) owns the rows k = kk + 16 j of four columns at a time, so that what
// depends on the row k of f alone is touched once per lane: x_k, f_k, the first-order and diagonal columns of direction i sit
// in registers (read straight from global: 128-byte runs, the same lines for the four column groups); per element there is
// one read of the block of first-order columns, one of the diagonal column of the second direction, at most one LDS read (the
// acceleration, rows k >= NV) and the store(s).
// DIAG_ROW = false: the diagonal second-order column (i, i) is read from d1g.  DIAG_ROW = true (torque-level rows): lane NV
// of the wave has evaluated the point "direction i alone" on the same cached operands (an otherwise idle lane); the column
// (problem.hpp:192-222: 2 ((f(x + eps e_i) - f(x)) - eps f_col_i) / eps^2) is formed here, kept for the row's own
// off-diagonal entries and written to d1g for the velocity- and configuration-level kernels that follow.
// out: the block [NV][n]; mirror (or null): column c at mirror + c * mstride; a2: contiguous [NV][n]; d2: column c at
// d2 + c * dstride; i: first direction (an x index); jb: x index of the second direction of column 0, or >= 2 NV for u directions
template <int NV, bool DIAG_ROW, class ST>
__device__ __forceinline__ void rowblock_emit(const ST& S, int lane, int i, int jb, double* __restrict__ out, double* __restrict__ mirror,
                                              int64_t mstride, const double* __restrict__ a2, const double* __restrict__ d2, int64_t dstride,
                                              double dt, const double* __restrict__ f0g, const double* __restrict__ a1g,
                                              double* __restrict__ d1g, const double* __restrict__ xg, bool skip_top = false) {
  constexpr int n = 2 * NV, RW = 16, NJ = (n + RW - 1) / RW, CG = LBS / RW, NIT = (NV + CG - 1) / CG;
  const int r1 = i % NV;                             // the configuration row the first direction (an x direction) touches
  const double eps = sqrt(sqrt(DBL_EPSILON));
  const double eps2 = eps * eps;
  const int kk = lane % RW, cg = lane / RW;
  double xk[NJ], xvk[NJ], f0k[NJ], a1k[NJ], d1k[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int k = kk + RW * j;
    const int kc = k < n ? k : n - 1;
    f0k[j] = f0g[kc];
    a1k[j] = a1g[kc];
    double x = xg[kc];
    if (kc == i) x = x + eps;
    xk[j] = x;
    double xv = xg[kc < NV ? NV + kc : kc];
    if (NV + kc == i) xv = xv + eps;
    xvk[j] = xv;
    if constexpr (DIAG_ROW) {
      // problem.hpp:192-222 for direction i alone: 2 ((f(x + eps e_i) - f(x)) - eps f_col_i) / eps^2; row NV of S.q is the
      // evaluation of that point (the wave's otherwise idle lane NV)
      double fv;
      if (kc < NV) {
        const double vo = dt * xv;
        fv = x + vo;
      } else {
        fv = x + S.q[NV * (NV + 1) + (kc - NV)] * dt;
      }
      double df = fv - f0k[j];
      df -= eps * a1k[j];
      df *= 2;
      const double dd = df / eps2;
      d1k[j] = dd;
      if (cg == 0 && k < n && !(skip_top && k < NV && k != r1)) d1g[k] = dd;   // (a configuration row other than r1: an exact zero, in memory)
    } else {
      d1k[j] = d1g[kc];
    }
  }
  // software-pipelined over the column groups: the operands of group it + 1 are requested before group it is formed and stored
  // (the evaluation's registers are dead here: room for a second operand buffer)
  double va2[2][NJ], vd2[2][NJ];
  auto fetch = [&](int it, double (&A)[NJ], double (&D)[NJ]) {
    const int c = it * CG + cg;
    const int cc = c < NV ? c : NV - 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = kk + RW * j;
      const int kc = k < n ? k : n - 1;
      A[j] = a2[kc + cc * n];
      D[j] = d2[kc + cc * dstride];
    }
  };
  auto form = [&](int it, const double (&A)[NJ], const double (&D)[NJ]) {
    const int c = it * CG + cg;
    const bool cv = c < NV;
    const int cc = cv ? c : NV - 1;
    const int jp = jb + cc;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = kk + RW * j;
      if (!(cv && k < n)) continue;
      // configuration rows other than those of the two directions: exact zeros, already in memory (LinParams::skip_top)
      if (skip_top && k < NV && k != r1 && !(jp < n && jp % NV == k)) continue;
      // row k of f(x + dx, u + du) (dynamics_t::eval_to, problem.hpp:441-461)
      double x = xk[j];
      if (k == jp) x = x + eps;
      double fv;
      if (k < NV) {
        double xv = xvk[j];
        if (NV + k == jp) xv = xv + eps;
        const double vo = dt * xv;
        fv = x + vo;
      } else {
        fv = x + S.q[cc * (NV + 1) + (k - NV)] * dt;
      }
      double df = fv - f0k[j];                          // difference_out
      df -= eps * a1k[j];
      df -= eps * A[j];
      df *= 2;
      const double val = 0.5 * (df / eps2 - d1k[j] - D[j]);
      out[k + c * n] = val;
      if (mirror) mirror[k + c * mstride] = val;
    }
  };
  if (DEV_NO_EMIT) return;
  fetch(0, va2[0], vd2[0]);
  for (int it = 0; it < NIT; it += 2) {
    if (it + 1 < NIT) fetch(it + 1, va2[1], vd2[1]);
    form(it, va2[0], vd2[0]);
    if (it + 1 < NIT) {
      if (it + 2 < NIT) fetch(it + 2, va2[0], vd2[0]);
      form(it + 1, va2[1], vd2[1]);
    }
  }
}

// ---- torque level: the (x_i, u_j) and (u_i, u_j) points -------------------------------------------------------------
// One wave per group; the groups of one (instance, t):
//   g <  nv        : (q_g, u_lane)       cfg 1+g, vcfg nv+1+g   38 of 64 lanes
//   g <  2 nv      : (v_{g-nv}, u_lane)  cfg 0,   vcfg 1+(g-nv) 38 of 64 lanes
//   g >= 2 nv      : (u_i, u_j) pairs    cfg 0,   vcfg 0        64 pairs per wave
template <class T, bool ROWS>
__global__ __launch_bounds__(LBS, ROWS ? 1 : 3) void lin_static_tau_kernel(LinParams p) {
  constexpr int nv = T::N, n = 2 * nv;
  constexpr int TRI = nv * (nv - 1) / 2, GU = (TRI + LBS - 1) / LBS, G = ROWS ? 2 * nv : GU;
  const int64_t bt = blockIdx.x / G;
  const int g = (int)(blockIdx.x % G) + (ROWS ? 0 : 2 * nv);
  const int lane = threadIdx.x;
  const int64_t Tn = p.d.T;
  const int b = (int)(bt / Tn);
  const int64_t t = bt % Tn;
  const DevModel& m = *p.model;
  int i, j, cfg, vcfg;
  bool valid;
  if (g < nv) { i = g; j = n + lane; cfg = 1 + g; vcfg = nv + 1 + g; valid = lane < nv; }
  else if (g < 2 * nv) { i = g; j = n + lane; cfg = 0; vcfg = 1 + (g - nv); valid = lane < nv; }
  else {
    const int pid = (g - 2 * nv) * LBS + lane;
    valid = pid < TRI;
    tri_index(valid ? pid : 0, nv, i, j);
    i += n; j += n; cfg = 0; vcfg = 0;
  }
  double eps = sqrt(sqrt(DBL_EPSILON));
  const double* __restrict__ qc = p.qcache + (bt * (nv + 1) + cfg) * (int64_t)nv * rbd::QC_STRIDE;
  const double* __restrict__ vc = p.vcache + (bt * (2 * nv + 1) + vcfg) * (int64_t)nv * rbd::VC_STRIDE;
  const double* __restrict__ xg = p.x + ((int64_t)b * (Tn + 1) + t) * n;
  const double* __restrict__ ug = p.u + ((int64_t)b * Tn + t) * nv;
  const int iu = i - n, ju = j - n;
  // rows: the staged operands are dead once the acceleration pass is over, the output stage's buffers are not alive
  // before: one LDS region for both (the evaluation and the output stage are a barrier apart)
  __shared__ union TauLds { double P[ROWS ? nv * PS + nv * rbd::VC_STRIDE : 1]; RowStage<nv> S; } s_lds;
  double* s_P = s_lds.P;
  // rows: the v-cache block of the row goes to LDS as well, in one coalesced pass (all of it in flight at once).  Through the
  // scalar path its 76 per-joint reads are 76 serialised L2 round trips per wave (measured: 19.5 -> 17.6 ms at 64 seeds)
  unsigned int w = 0;
  double* s_V = s_P + nv * PS;
  constexpr int KH = ROWS ? stage_split<T>() : 0;
  StageRegs<nv, 0, (ROWS && KH > 0) ? KH : 1> lower;   // (KH == 0: joint 0 once more, harmless)            // joints 0 .. KH-1: in flight through the first half of the leaf -> root pass
  if constexpr (ROWS) {
    StageRegs<nv, KH, nv> upper;
    upper.load(qc, vc, lane);
    lower.load(qc, vc, lane);
    upper.park(s_P, s_V, lane);
    rbd::coop_sync<true>();                         // one wave per workgroup: LDS is in order, no vmcnt(0) behind this fence
  } else {
    w = warm_block<nv * rbd::VC_STRIDE * 8>(vc, lane);
    w ^= warm_block<nv * rbd::QC_STRIDE * 8>(qc, lane);
  }
  if (w == 0x7fc01234u) eps = 0.0;     // never true for cache contents that are finite doubles in practice; orders the chain
  TauState<T> s;
  auto tau = [&](int k) { double v = ug[k]; if (k == iu) v = v + eps; if (k == ju) v = v + eps; return v; };
  if constexpr (ROWS) {
    // joints nv-1 .. KH+1 only touch records KH .. nv-1 (their own and their parents': checked below), joint KH needs its parent's
    static_assert(!ROWS || parents_at_least<T>(KH + 1, KH), "the first half of the pass must not reach below record KH");
    tau_up_range<T, PS, nv - 1>(m, s_P, s_V, tau, s, std::make_integer_sequence<int, nv - 1 - KH>{});
    lower.park(s_P, s_V, lane);
    rbd::coop_sync<true>();
    tau_up_range<T, PS, KH>(m, s_P, s_V, tau, s, std::make_integer_sequence<int, KH + 1>{});
    tau_down_all<T, PS>(m, s_P, s_V, s, std::make_integer_sequence<int, nv>{});
  } else {
    tau_up_all<T, rbd::QC_STRIDE>(m, qc, vc, tau, s, std::make_integer_sequence<int, nv>{});
    tau_down_all<T, rbd::QC_STRIDE>(m, qc, vc, s, std::make_integer_sequence<int, nv>{});
  }
  // The output stage reads its pointers from the kernel-argument segment only now: taken from `p` they would be
  // loaded at kernel entry and stay live through the whole evaluation, and the scalar registers would spill.
  __builtin_amdgcn_sched_barrier(0);
  typedef __attribute__((address_space(4))) const LinParams* kernarg_t;
  const kernarg_t kp = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (ROWS) {
    RowStage<nv>& S = s_lds.S;
    const int mm = nv;
    __syncthreads();                       // every lane is done with the staged operands
    // unconditional on purpose: under `if (valid)` the optimiser sinks the whole evaluation into the branch, away from
    // its operand loads, and every operand then spills
    const int row = lane <= nv ? lane : nv + 1;      // lane nv: direction i alone (its second direction index falls off the controls)
#pragma unroll
    for (int k = 0; k < nv; ++k) S.q[row * (nv + 1) + k] = s.uu[k];
    // (the staging loop comes after these stores: the evaluation must meet its first use in its own basic block)
    __syncthreads();     // row nv of q is another lane's
    // column (k, u_c) of the (x_i, u) slab of f_ux: k + c n + i n m
    rowblock_emit<nv, true>(S, lane, i, n, kp->fux + bt * n * mm * n + (int64_t)i * n * mm, nullptr, 0, kp->fu + bt * n * mm,
                            kp->fuu + bt * n * mm * mm, (int64_t)n + (int64_t)n * mm, m.dt, kp->f_val + bt * n,
                            kp->fx + bt * n * n + (int64_t)i * n, kp->fxx + bt * n * n * n + (int64_t)i * n + (int64_t)i * n * n, xg, kp->skip_top != 0);
  } else {
    __shared__ OutStage<nv, LBS / 2> S;
    LinParams po;
    po.f_val = kp->f_val; po.fx = kp->fx; po.fu = kp->fu; po.fxx = kp->fxx; po.fux = kp->fux; po.fuu = kp->fuu; po.skip_qv_mirror = kp->skip_qv_mirror; po.skip_top = kp->skip_top;
    const double dt = m.dt;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      if (pass) __syncthreads();                  // the first half's slots have been read
      if (lane / (LBS / 2) == pass) {
#pragma unroll
        for (int k = 0; k < nv; ++k) S.qdd[(lane % (LBS / 2)) * (nv + 1) + k] = s.uu[k];
      }
      offdiag_emit<nv, LBS / 2>(po, S, valid, i, j, bt, xg, dt, pass);
    }
  }
}

// ---- velocity level: the (q_i, v_j) and (v_i, v_j) points ----------------------------------------------------------
// The evaluation is rbd::aba_vu_cached (velocity pass, bias-force pass, acceleration pass on the cached q-part), but
// its first two passes are walked chain by chain (a chain = a run of single-child joints): down a chain computing the
// link velocities, back up it forming the bias forces, so that only one chain's velocities are alive at a time; the
// velocity of a branching joint is formed when its first child chain needs it.  The third pass recomputes the link
// velocities on its way down instead of keeping 12 doubles per joint from the first.  u_i lives in LDS (the buffer the
// output stage reads the accelerations from).
template <class T> constexpr int n_children(int k) {
  int c = 0;
  for (int j = k + 1; j < T::N; ++j) if (T::parent[j] == k) ++c;
  return c;
}
template <class T> constexpr bool chain_start(int k) { return T::parent[k] < 0 || T::parent[k] != k - 1 || n_children<T>(T::parent[k]) > 1; }
template <class T> constexpr bool chain_end(int k) { return k == T::N - 1 || chain_start<T>(k + 1); }
template <class T> constexpr int chain_first(int k) { while (!chain_start<T>(k)) --k; return k; }
// longest chain of the tree: sizes the LDS buffer of the chain-wise sweeps (8 for the Talos-like tree: its arms)
template <class T> constexpr int max_chain() {
  int best = 1;
  for (int k = 0; k < T::N; ++k)
    if (chain_end<T>(k)) { const int len = k - chain_first<T>(k) + 1; if (len > best) best = len; }
  return best;
}

template <class T>
struct VelState {
  double vel[T::N][6];   // link velocities (a chain at a time, plus the branching joints)
  double acc[T::N][6];   // bias-force accumulators, then link accelerations
  double uu[T::N];       // u_i, then the joint acceleration (contexts with UQS == 0; else they live in LDS behind uq)
};

template <int UQS_, int QS_ = rbd::QC_STRIDE>
struct VelCtx {
  static constexpr int UQS = UQS_;   // stride between consecutive joints of this lane's u_i / acceleration buffer in LDS;
                                     // 0: no buffer, the values stay in registers (VelState::uu)
  static constexpr int QS = QS_;     // doubles per joint of the (E | r | U | 1/D) block behind qp
  const DevModel* m;
  const double* qp;                  // (E | r | U | 1/D): the q-cache block itself, or its copy in LDS
  const double* __restrict__ qc;
  const double* __restrict__ xg;
  const double* __restrict__ ug;
  double* uq;            // LDS: u_i, then the joint acceleration, of this lane (joint K at uq[K * UQS])
  int i, j;              // perturbed directions (x indices; q directions do not change v)
  double eps;
};

template <int NV, class C>
__device__ __forceinline__ double lane_v(const C& c, int K) {
  double v = c.xg[NV + K];
  if (NV + K == c.i) v = v + c.eps;
  if (NV + K == c.j) v = v + c.eps;
  return v;
}

template <class T, int K, class C>
__device__ __forceinline__ void joint_vel(const C& c, const double* vel_par, double* vel) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* E = c.qp + K * C::QS;
  const double* r = E + 9;
  const double* a = c.m->axis[K];
  const double vK = lane_v<T::N>(c, K);
  double vJ[6] = {0, 0, 0, 0, 0, 0};
  vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
  if constexpr (T::parent[K] >= 0) rbd::xform_motion(E, r, vel_par, vel);
  else {
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
}
// velocity of joint K from the root (used for the branching joints)
template <class T, int K, class C>
__device__ __forceinline__ void vel_from_root(const C& c, double* vel) {
  if constexpr (T::parent[K] >= 0) {
    double vp[6];
    vel_from_root<T, T::parent[K]>(c, vp);
    joint_vel<T, K>(c, vp, vel);
  } else {
    joint_vel<T, K>(c, nullptr, vel);
  }
}
template <class T, int K, class C>
__device__ __forceinline__ void bias_force0(const C& c, const double* vel, double* pA) {
  double Iv[6];
  rbd::sym6_mv(c.m->I6[K], vel, Iv);
  rbd::crf(vel, Iv, pA);
}

template <class T, int K, int E, class C>
__device__ __forceinline__ void chain_down(const C& c, VelState<T>& s) {
  if constexpr (T::parent[K] >= 0) joint_vel<T, K>(c, s.vel[T::parent[K]], s.vel[K]);
  else joint_vel<T, K>(c, nullptr, s.vel[K]);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K < E) chain_down<T, K + 1, E>(c, s);
}

template <class T, int K, int F, class C>
__device__ __forceinline__ void chain_up(const C& c, VelState<T>& s) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* E = c.qp + K * C::QS;
  const double* r = E + 9;
  const double* U = E + 12;
  const double dinv = E[18];
  const double* Ia = c.qc + K * rbd::QC_STRIDE + 19;
  const double* a = c.m->axis[K];
  double pAi[6];
  if constexpr (has_child<T>(K)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) pAi[k] = s.acc[K][k];
  } else {
    bias_force0<T, K>(c, s.vel[K], pAi);
  }
  double sp = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) sp += a[k] * pAi[o + k];
  const double ui = c.ug[K] - sp;
  if constexpr (C::UQS == 0) s.uu[K] = ui; else c.uq[K * C::UQS] = ui;
  constexpr int par = T::parent[K];
  if constexpr (par >= 0) {
    const double vK = lane_v<T::N>(c, K);
    double vJ[6] = {0, 0, 0, 0, 0, 0}, cb[6], pa[6], Iac[6], fp[6];
    vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
    rbd::crm(s.vel[K], vJ, cb);
    rbd::sym6_mv(Ia, cb, Iac);
#pragma unroll
    for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
    rbd::xform_force_T(E, r, pa, fp);
    if constexpr (first_contrib<T>(K)) {
      double p0[6];
      bias_force0<T, par>(c, s.vel[par], p0);
#pragma unroll
      for (int k = 0; k < 6; ++k) s.acc[par][k] = p0[k] + fp[k];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) s.acc[par][k] += fp[k];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K > F) chain_up<T, K - 1, F>(c, s);
}

// chains in descending order of their joints: a chain is processed when the fold reaches its last joint
template <class T, int K, class C>
__device__ __forceinline__ void chain_at(const C& c, VelState<T>& s) {
  if constexpr (chain_end<T>(K)) {
    constexpr int F = chain_first<T>(K), P = T::parent[F];
    if constexpr (P >= 0) {
      // the largest-index child chain of a branching joint is the first to need (and to form) that joint's velocity
      if constexpr (first_contrib<T>(F) && n_children<T>(P) > 1) vel_from_root<T, P>(c, s.vel[P]);
    }
    chain_down<T, F, K>(c, s);
    chain_up<T, K, F>(c, s);
  }
}
template <class T, class C, int... Ks>
__device__ __forceinline__ void vel_up_all(const C& c, VelState<T>& s, std::integer_sequence<int, Ks...>) {
  (chain_at<T, T::N - 1 - Ks>(c, s), ...);
}

// acceleration pass (rbd::aba_vu_cached, third loop), link velocities recomputed on the way
template <class T, int K, class C>
__device__ __forceinline__ void vel_down(const C& c, VelState<T>& s) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  constexpr int par = T::parent[K];
  const double* E = c.qp + K * C::QS;
  const double* r = E + 9;
  const double* U = E + 12;
  const double dinv = E[18];
  const double* a = c.m->axis[K];
  const double vK = lane_v<T::N>(c, K);
  double vJ[6] = {0, 0, 0, 0, 0, 0}, vel[6], cb[6], ap[6];
  vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
  if constexpr (par >= 0) {
    rbd::xform_motion(E, r, s.vel[par], vel);
    rbd::xform_motion(E, r, s.acc[par], ap);
  } else {
    const double a0[6] = {0, 0, 0, -c.m->gravity[0], -c.m->gravity[1], -c.m->gravity[2]};
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] = 0.0;
    rbd::xform_motion(E, r, a0, ap);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
  rbd::crm(vel, vJ, cb);
  double sum = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; sum += U[k] * ap[k]; }
  double ui;
  if constexpr (C::UQS == 0) ui = s.uu[K]; else ui = c.uq[K * C::UQS];
  const double qd = (ui - sum) * dinv;
  if constexpr (C::UQS == 0) s.uu[K] = qd; else c.uq[K * C::UQS] = qd;
  if constexpr (has_child<T>(K)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { s.vel[K][k] = vel[k]; s.acc[K][k] = ap[k]; }
    s.acc[K][o] += a[0] * qd; s.acc[K][o + 1] += a[1] * qd; s.acc[K][o + 2] += a[2] * qd;
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <class T, class C, int... Ks>
__device__ __forceinline__ void vel_down_all(const C& c, VelState<T>& s, std::integer_sequence<int, Ks...>) {
  (vel_down<T, Ks>(c, s), ...);
}

// One wave per group; the groups of one (instance, t):
//   ROWS:  g < nv : (q_g, v_lane)      cfg 1+g   38 of 64 lanes, row-block output (+ mirror image)
//   !ROWS: (v_i, v_j) pairs, cfg 0, 64 pairs per wave, generic output
template <class T, bool ROWS>
__global__ __launch_bounds__(LBS, ROWS ? 3 : 1) void lin_static_vel_kernel(LinParams p) {
  constexpr int nv = T::N, n = 2 * nv;
  constexpr int TRI = nv * (nv - 1) / 2, GU = (TRI + LBS - 1) / LBS, G = ROWS ? nv : GU;
  const int64_t bt = blockIdx.x / G;
  const int g = (int)(blockIdx.x % G);
  const int lane = threadIdx.x;
  const int64_t Tn = p.d.T;
  const int b = (int)(bt / Tn);
  const int64_t t = bt % Tn;
  int i, j, cfg;
  bool valid;
  if (ROWS) { i = g; j = nv + lane; cfg = 1 + g; valid = lane < nv; }
  else {
    const int pid = g * LBS + lane;
    valid = pid < TRI;
    tri_index(valid ? pid : 0, nv, i, j);
    i += nv; j += nv; cfg = 0;
  }
  // rows: u_i / the accelerations stay in registers, and the staged operands (dead after the acceleration pass) share
  // their LDS with the output stage's buffers (alive after it)
  using Stage = typename std::conditional<ROWS, RowStage<nv>, OutStage<nv>>::type;
  // rows: the whole q-cache block of the row (placements, U, 1/D and the articulated inertias: 12 KB) is staged in LDS, all of
  // its loads in flight at once; nothing of the evaluation goes through the scalar path's serialised round trips
  __shared__ union VelLds { Stage S; double P[ROWS ? nv * rbd::QC_STRIDE : 1]; } s_lds;
  Stage& S = s_lds.S;
  double* s_P = s_lds.P;
  VelCtx<ROWS ? 0 : 1, rbd::QC_STRIDE> c;
  c.m = p.model;
  c.qc = p.qcache + (bt * (nv + 1) + cfg) * (int64_t)nv * rbd::QC_STRIDE;
  c.qp = ROWS ? s_P : c.qc;
  c.xg = p.x + ((int64_t)b * (Tn + 1) + t) * n;
  c.ug = p.u + ((int64_t)b * Tn + t) * nv;
  c.i = i; c.j = j;
  c.eps = sqrt(sqrt(DBL_EPSILON));
  // pairs: this lane's row of the output stage's acceleration buffer (lane-major, odd stride)
  if constexpr (ROWS) c.uq = nullptr;
  else c.uq = &S.qdd[lane * (nv + 1)];
  if constexpr (ROWS) {
    constexpr int NW = nv * rbd::QC_STRIDE, CW = (NW + LBS - 1) / LBS;
    double rq[CW];
#pragma unroll
    for (int r = 0; r < CW; ++r) { const int idx = r * LBS + lane; rq[r] = c.qc[idx < NW ? idx : NW - 1]; }
#pragma unroll
    for (int r = 0; r < CW; ++r) { const int idx = r * LBS + lane; s_P[idx < NW ? idx : NW - 1] = rq[r]; }
    c.qc = s_P;
    rbd::coop_sync<true>();
  } else {
    const unsigned int w = warm_block<nv * rbd::QC_STRIDE * 8>(c.qc, lane);
    if (w == 0x7fc01234u) c.eps = 0.0;     // never true in practice; orders the evaluation behind the warm-up
  }
  VelState<T> s;
  vel_up_all<T>(c, s, std::make_integer_sequence<int, nv>{});
  vel_down_all<T>(c, s, std::make_integer_sequence<int, nv>{});
  __builtin_amdgcn_sched_barrier(0);
  typedef __attribute__((address_space(4))) const LinParams* kernarg_t;
  const kernarg_t kp = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  const double dt = c.m->dt;
  if constexpr (ROWS) {
    double* fxx = kp->fxx + bt * n * n * n;
    const double* fxb = kp->fx + bt * n * n;
    __syncthreads();                       // every lane is done with the staged operands
    // lane-major with an odd stride (the layout the row-block output reads), idle lanes share the spare row; unconditional
    // stores right behind the evaluation (see the torque-level kernel)
    const int row = valid ? lane : nv;
#pragma unroll
    for (int k = 0; k < nv; ++k) S.q[row * (nv + 1) + k] = s.uu[k];
    __syncthreads();
    // column (k, v_c) of slab q_i of f_xx: k + (nv + c) n + i n n; its mirror image: column q_i of slab v_c
    rowblock_emit<nv, false>(S, lane, i, nv, fxx + (int64_t)nv * n + (int64_t)i * n * n,
                             kp->skip_qv_mirror ? nullptr : fxx + (int64_t)i * n + (int64_t)nv * n * n, (int64_t)n * n,
                             fxb + (int64_t)nv * n, fxx + (int64_t)nv * n + (int64_t)nv * n * n, (int64_t)n + (int64_t)n * n, dt,
                             kp->f_val + bt * n, fxb + (int64_t)i * n, fxx + (int64_t)i * n + (int64_t)i * n * n, c.xg, kp->skip_top != 0);
  } else {
    LinParams po;
    po.f_val = kp->f_val; po.fx = kp->fx; po.fu = kp->fu; po.fxx = kp->fxx; po.fux = kp->fux; po.fuu = kp->fuu; po.skip_qv_mirror = kp->skip_qv_mirror; po.skip_top = kp->skip_top;
    offdiag_emit<nv, LBS>(po, S, valid, i, j, bt, c.xg, dt);
  }
}

// ---- configuration level: the (q_i, q_j) points ---------------------------------------------------------------------
// Every point has its own configuration, so the whole articulated-body algorithm runs per lane (rbd::aba_tree): the
// chain-wise sweep of the velocity level, with the articulated inertias accumulated alongside the bias forces.  Only
// the placements of joints i and j differ from the base configuration: each lane fetches those two from the q-cache of
// configurations 1+i and 1+j once, everything else comes through the scalar path.  What the third pass needs from the
// second (U, 1/D, u: 8 doubles per joint) does not fit in registers or LDS; it goes through a per-wave workspace,
// written and read back once, 512-byte coalesced rows.
constexpr int WS_PER_JOINT = 8;

template <class T>
struct CfgState {
  double vel[T::N][6];
  double accP[T::N][6];
  double accI[T::N][21];
  double w[2][WS_PER_JOINT];   // third pass: workspace values of the current / next joint
};

struct CfgCtx {
  const DevModel* __restrict__ m;
  const double* __restrict__ qc0;   // q-cache of the base configuration
  const double* __restrict__ xg;
  const double* __restrict__ ug;
  double* __restrict__ W;           // this lane's column of the wave's workspace: (K, e) at W[(K * 8 + e) * LBS]
  double* lvel;                     // LDS: link velocities of the chain being swept, (slot, k) at lvel[(slot * 6 + k) * LBS]
  int i, j;
  const double* own;                // null: joints i, j take E | r from configurations 1+i, 1+j of the q-cache;
                                    // else this lane's own E | r of joint i (first order: a step the cache does not hold)
};

template <int K, int NV>
__device__ __forceinline__ void lane_placement(const CfgCtx& c, double* P) {
  // E | r of joint K at q_K and at q_K + eps (configuration 1+K of the q-cache): both wave-uniform, selected per lane
  const double* base = c.qc0 + K * rbd::QC_STRIDE;
  const double* pert = c.own ? c.own : c.qc0 + ((int64_t)(1 + K) * NV + K) * rbd::QC_STRIDE;
  const bool own = K == c.i || K == c.j;
#pragma unroll
  for (int e = 0; e < 12; ++e) P[e] = own ? pert[e] : base[e];
}

template <class T, int K>
__device__ __forceinline__ void cfg_joint_vel(const CfgCtx& c, const double* P, const double* vel_par, double* vel) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* a = c.m->axis[K];
  const double vK = c.xg[T::N + K];
  double vJ[6] = {0, 0, 0, 0, 0, 0};
  vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
  if constexpr (T::parent[K] >= 0) rbd::xform_motion(P, P + 9, vel_par, vel);
  else {
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
}
template <class T, int K>
__device__ __forceinline__ void cfg_vel_from_root(const CfgCtx& c, double* vel) {
  double P[12];
  lane_placement<K, T::N>(c, P);
  if constexpr (T::parent[K] >= 0) {
    double vp[6];
    cfg_vel_from_root<T, T::parent[K]>(c, vp);
    cfg_joint_vel<T, K>(c, P, vp, vel);
  } else {
    cfg_joint_vel<T, K>(c, P, nullptr, vel);
  }
}
template <int K>
__device__ __forceinline__ void cfg_bias_force0(const CfgCtx& c, const double* vel, double* pA) {
  double Iv[6];
  rbd::sym6_mv(c.m->I6[K], vel, Iv);
  rbd::crf(vel, Iv, pA);
}

// the link velocities of a chain go to LDS slot (K - F); `cur` carries the velocity of the previous joint
template <class T, int K, int F, int E>
__device__ __forceinline__ void cfg_chain_down(const CfgCtx& c, CfgState<T>& s, double* cur) {
  double P[12], vel[6];
  lane_placement<K, T::N>(c, P);
  if constexpr (T::parent[K] >= 0) cfg_joint_vel<T, K>(c, P, cur, vel);
  else cfg_joint_vel<T, K>(c, P, nullptr, vel);
#pragma unroll
  for (int k = 0; k < 6; ++k) { cur[k] = vel[k]; c.lvel[((K - F) * 6 + k) * LBS] = vel[k]; }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K < E) cfg_chain_down<T, K + 1, F, E>(c, s, cur);
}
template <class T, int K, int F>
__device__ __forceinline__ void chain_vel_load(const CfgCtx& c, const CfgState<T>& s, double* vel) {
  if constexpr (K >= F) {
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] = c.lvel[((K - F) * 6 + k) * LBS];
  } else {   // the joint the chain hangs from: a branching joint, kept in registers
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] = s.vel[K][k];
  }
}

// leaf -> root step of joint K (rbd::aba_tree, second loop)
template <class T, int K, int F>
__device__ __forceinline__ void cfg_chain_up(const CfgCtx& c, CfgState<T>& s) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* a = c.m->axis[K];
  double IA[21], pAi[6], U[6], velK[6];
  chain_vel_load<T, K, F>(c, s, velK);
  if constexpr (has_child<T>(K)) {
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] = s.accI[K][k];
#pragma unroll
    for (int k = 0; k < 6; ++k) pAi[k] = s.accP[K][k];
  } else {
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] = c.m->I6[K][k];
    cfg_bias_force0<K>(c, velK, pAi);
  }
  double d = 0, sp = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) U[r] = IA[rbd::sidx(r, o)] * a[0] + IA[rbd::sidx(r, o + 1)] * a[1] + IA[rbd::sidx(r, o + 2)] * a[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) { d += a[k] * U[o + k]; sp += a[k] * pAi[o + k]; }
  const double dinv = 1.0 / d;
  const double ui = c.ug[K] - sp;
#pragma unroll
  for (int k = 0; k < 6; ++k) c.W[(K * WS_PER_JOINT + k) * LBS] = U[k];
  c.W[(K * WS_PER_JOINT + 6) * LBS] = dinv;
  c.W[(K * WS_PER_JOINT + 7) * LBS] = ui;
  constexpr int par = T::parent[K];
  if constexpr (par >= 0) {
    double P[12], Ia[21], vJ[6] = {0, 0, 0, 0, 0, 0}, cb[6], pa[6], Iac[6], fp[6];
    lane_placement<K, T::N>(c, P);
    const double vK = c.xg[T::N + K];
    vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = 0; cc <= r; ++cc) Ia[rbd::sidx(r, cc)] = IA[rbd::sidx(r, cc)] - U[r] * U[cc] * dinv;
    rbd::crm(velK, vJ, cb);
    rbd::sym6_mv(Ia, cb, Iac);
#pragma unroll
    for (int k = 0; k < 6; ++k) pa[k] = pAi[k] + Iac[k] + U[k] * (ui * dinv);
    if constexpr (first_contrib<T>(K)) {
      double velP[6];
      chain_vel_load<T, par, F>(c, s, velP);
#pragma unroll
      for (int k = 0; k < 21; ++k) s.accI[par][k] = c.m->I6[par][k];
      cfg_bias_force0<par>(c, velP, s.accP[par]);
    }
    rbd::add_xtix(P, P + 9, Ia, s.accI[par]);
    rbd::xform_force_T(P, P + 9, pa, fp);
#pragma unroll
    for (int k = 0; k < 6; ++k) s.accP[par][k] += fp[k];
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K > F) cfg_chain_up<T, K - 1, F>(c, s);
}

template <class T, int K>
__device__ __forceinline__ void cfg_chain_at(const CfgCtx& c, CfgState<T>& s) {
  if constexpr (chain_end<T>(K)) {
    constexpr int F = chain_first<T>(K), P = T::parent[F];
    if constexpr (P >= 0) {
      if constexpr (first_contrib<T>(F) && n_children<T>(P) > 1) cfg_vel_from_root<T, P>(c, s.vel[P]);
    }
    double cur[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (P >= 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) cur[k] = s.vel[P][k];
    }
    cfg_chain_down<T, F, F, K>(c, s, cur);
    cfg_chain_up<T, K, F>(c, s);
  }
}
template <class T, int... Ks>
__device__ __forceinline__ void cfg_up_all(const CfgCtx& c, CfgState<T>& s, std::integer_sequence<int, Ks...>) {
  (cfg_chain_at<T, T::N - 1 - Ks>(c, s), ...);
}

// acceleration pass (rbd::aba_tree, third loop); link velocities recomputed, (U, 1/D, u) read back one joint ahead
// where the acceleration pass leaves joint K's acceleration: uq[K * STRIDE] (STRIDE = LBS: [joint][lane] rows of the
// pair output stage; STRIDE = 1: this lane's own row)
template <int STRIDE_>
struct CfgDownOut { double* uq; static constexpr int STRIDE = STRIDE_; };

template <class T, int K, class O>
__device__ __forceinline__ void cfg_down(const CfgCtx& c, CfgState<T>& s, const O& out) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  constexpr int par = T::parent[K];
  if constexpr (K + 1 < T::N) {
#pragma unroll
    for (int e = 0; e < WS_PER_JOINT; ++e) s.w[(K + 1) & 1][e] = c.W[((K + 1) * WS_PER_JOINT + e) * LBS];
  }
  const double* U = s.w[K & 1];
  const double dinv = s.w[K & 1][6], ui = s.w[K & 1][7];
  const double* a = c.m->axis[K];
  const double vK = c.xg[T::N + K];
  double P[12], vJ[6] = {0, 0, 0, 0, 0, 0}, vel[6], cb[6], ap[6];
  lane_placement<K, T::N>(c, P);
  vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
  if constexpr (par >= 0) {
    rbd::xform_motion(P, P + 9, s.vel[par], vel);
    rbd::xform_motion(P, P + 9, s.accP[par], ap);
  } else {
    const double a0[6] = {0, 0, 0, -c.m->gravity[0], -c.m->gravity[1], -c.m->gravity[2]};
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] = 0.0;
    rbd::xform_motion(P, P + 9, a0, ap);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
  rbd::crm(vel, vJ, cb);
  double sum = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) { ap[k] += cb[k]; sum += U[k] * ap[k]; }
  const double qd = (ui - sum) * dinv;
  out.uq[K * (O::STRIDE == LBS ? LBS : 1)] = qd;
  if constexpr (has_child<T>(K)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { s.vel[K][k] = vel[k]; s.accP[K][k] = ap[k]; }
    s.accP[K][o] += a[0] * qd; s.accP[K][o + 1] += a[1] * qd; s.accP[K][o + 2] += a[2] * qd;
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <class T, class O, int... Ks>
__device__ __forceinline__ void cfg_down_all(const CfgCtx& c, CfgState<T>& s, const O& out, std::integer_sequence<int, Ks...>) {
  (cfg_down<T, Ks>(c, s, out), ...);
}

// One wave = 64 (q_i, q_j) pairs, i < j, in two kernels: the fused velocity / articulated-inertia / bias-force sweep
// (register hungry: one wave per SIMD), then the acceleration pass and the output stage (light: several waves per
// SIMD hide the workspace reads).  The pointers the evaluation reads through are separate restrict-qualified arguments:
// the workspace stores must not turn the scalar operand loads that follow them into per-lane loads.
template <class T>
__device__ __forceinline__ void cfg_setup(CfgCtx& c, const LinParams& p, const DevModel* __restrict__ model, const double* __restrict__ qcache,
                                          const double* __restrict__ xs, const double* __restrict__ us, double* __restrict__ ws,
                                          int64_t bt0, int64_t& bt, bool& valid) {
  constexpr int nv = T::N, n = 2 * nv;
  constexpr int TRI = nv * (nv - 1) / 2, GU = (TRI + LBS - 1) / LBS;
  bt = bt0 + blockIdx.x / GU;
  const int g = (int)(blockIdx.x % GU);
  const int lane = threadIdx.x;
  const int64_t Tn = p.d.T;
  const int b = (int)(bt / Tn);
  const int64_t t = bt % Tn;
  const int pid = g * LBS + lane;
  valid = pid < TRI;
  tri_index(valid ? pid : 0, nv, c.i, c.j);
  c.m = model;
  c.qc0 = qcache + (bt * (nv + 1)) * (int64_t)nv * rbd::QC_STRIDE;
  c.xg = xs + ((int64_t)b * (Tn + 1) + t) * n;
  c.ug = us + ((int64_t)b * Tn + t) * nv;
  c.W = ws + (int64_t)blockIdx.x * (nv * WS_PER_JOINT * LBS) + lane;
  c.own = nullptr;
}

template <class T>
__global__ __launch_bounds__(LBS) void lin_static_cfg_up_kernel(LinParams p, const DevModel* __restrict__ model, const double* __restrict__ qcache,
                                                                const double* __restrict__ xs, const double* __restrict__ us,
                                                                double* __restrict__ ws, int64_t bt0) {
  constexpr int nv = T::N;
  constexpr int MAXCH = max_chain<T>();  // longest chain of the topology
  __shared__ double s_vel[MAXCH * 6 * LBS];
  CfgCtx c;
  int64_t bt;
  bool valid;
  cfg_setup<T>(c, p, model, qcache, xs, us, ws, bt0, bt, valid);
  c.lvel = s_vel + threadIdx.x;
  CfgState<T> s;
  cfg_up_all<T>(c, s, std::make_integer_sequence<int, nv>{});
}

template <class T>
__global__ __launch_bounds__(LBS) void lin_static_cfg_down_kernel(LinParams p, const DevModel* __restrict__ model, const double* __restrict__ qcache,
                                                                  const double* __restrict__ xs, const double* __restrict__ us,
                                                                  double* __restrict__ ws, int64_t bt0) {
  constexpr int nv = T::N;
  __shared__ OutStage<nv> S;
  CfgCtx c;
  int64_t bt;
  bool valid;
  cfg_setup<T>(c, p, model, qcache, xs, us, ws, bt0, bt, valid);
  c.lvel = nullptr;
  CfgState<T> s;
#pragma unroll
  for (int e = 0; e < WS_PER_JOINT; ++e) s.w[0][e] = c.W[e * LBS];
  CfgDownOut<nv + 1> o{&S.qdd[threadIdx.x * (nv + 1)]};
  cfg_down_all<T>(c, s, o, std::make_integer_sequence<int, nv>{});
  __builtin_amdgcn_sched_barrier(0);
  typedef __attribute__((address_space(4))) const LinParams* kernarg_t;
  const kernarg_t kp = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  LinParams po;
  po.f_val = kp->f_val; po.fx = kp->fx; po.fu = kp->fu; po.fxx = kp->fxx; po.fux = kp->fux; po.fuu = kp->fuu; po.skip_qv_mirror = kp->skip_qv_mirror; po.skip_top = kp->skip_top;
  offdiag_emit<nv, LBS>(po, S, valid, c.i, c.j, bt, c.xg, model->dt);
}

// ---- first order: forward differences of f (problem.hpp:105-126 stepping, eps = sqrt(DBL_EPSILON)) -------------------
// Three waves per (instance, t): lane d < nv perturbs q_d (LEVEL 1, full evaluation with its own placement of joint
// d), v_d (LEVEL 2, velocity level on the base configuration) or u_d (LEVEL 3, torque level on the base (q, v)).
// Column d of f_x / f_u is (f(x + eps e_d) - f(x)) / eps; the nv columns of a wave are one contiguous block.
// DIAG: the same single-direction evaluations with eps = eps_mach^(1/4) give the diagonal second-order entries
// (problem.hpp:192-222): column (d, d) of the tensor is 2 ((f(x + eps e_d) - f(x)) - eps f_col_d) / eps^2; `out` then
// points at column (0, 0) of the level's tensor block, consecutive columns ostride apart, fcol at the level's first-order block
template <int NV, bool DIAG>
__device__ __forceinline__ void first_output(const double* q /* LDS, lane-major, stride NV + 1 */, int lane, int d0, double* __restrict__ out,
                                             int64_t ostride, const double* __restrict__ fcol, const double* __restrict__ f0,
                                             const double* __restrict__ xg, double dt, double eps, bool skip_top = false) {
  constexpr int n = 2 * NV, TOT = NV * n;
  const double eps2 = eps * eps;
  for (int e = lane; e < TOT; e += LBS) {
    const int c = e / n, k = e - c * n;
    const int d = d0 + c;                        // perturbed direction of this column (an x index, or >= n for u)
    if (DIAG && skip_top && k < NV && !(d < n && d % NV == k)) continue;   // configuration rows: exact zeros, in memory (LinParams::skip_top)
    double xk = xg[k];
    if (k == d) xk = xk + eps;
    double fv;
    if (k < NV) {
      double xv = xg[NV + k];
      if (NV + k == d) xv = xv + eps;
      const double vo = dt * xv;
      fv = xk + vo;
    } else {
      fv = xk + q[c * (NV + 1) + (k - NV)] * dt;
    }
    if constexpr (DIAG) {
      double df = fv - f0[k];
      df -= eps * fcol[e];
      df *= 2;
      out[k + c * ostride] = df / eps2;
    } else {
      out[e] = (fv - f0[k]) / eps;
    }
  }
}

template <class T, int LEVEL, bool DIAG>
__global__ __launch_bounds__(LBS) void lin_static_first_kernel(LinParams p, const DevModel* __restrict__ model, const double* __restrict__ qcache,
                                                               const double* __restrict__ xs, const double* __restrict__ us,
                                                               double* __restrict__ ws, int64_t bt0) {
  constexpr int nv = T::N, n = 2 * nv;
  const int64_t bt = bt0 + blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t Tn = p.d.T;
  const int b = (int)(bt / Tn);
  const int64_t t = bt % Tn;
  const bool valid = lane < nv;
  const double eps = DIAG ? sqrt(sqrt(DBL_EPSILON)) : sqrt(DBL_EPSILON);
  const double* __restrict__ qc0 = qcache + (bt * p.ncfg) * (int64_t)nv * rbd::QC_STRIDE;
  const double* __restrict__ xg = xs + ((int64_t)b * (Tn + 1) + t) * n;
  const double* __restrict__ ug = us + ((int64_t)b * Tn + t) * nv;
  __shared__ double s_q[(nv + 1) * (nv + 1)];
  double* uq = s_q + (valid ? lane : nv) * (nv + 1);
  if constexpr (LEVEL == 3) {
    const double* __restrict__ vc = p.vcache + (bt * p.nvcfg) * (int64_t)nv * rbd::VC_STRIDE;
    TauState<T> s;
    auto tau = [&](int k) { double v = ug[k]; if (k == lane) v = v + eps; return v; };
    tau_up_all<T, rbd::QC_STRIDE>(*model, qc0, vc, tau, s, std::make_integer_sequence<int, nv>{});
    tau_down_all<T, rbd::QC_STRIDE>(*model, qc0, vc, s, std::make_integer_sequence<int, nv>{});
#pragma unroll
    for (int k = 0; k < nv; ++k) uq[k] = s.uu[k];
  } else if constexpr (LEVEL == 2) {
    VelCtx<1> c;
    c.m = model; c.qc = qc0; c.qp = qc0; c.xg = xg; c.ug = ug; c.uq = uq;
    c.i = nv + lane; c.j = -1; c.eps = eps;
    VelState<T> s;
    vel_up_all<T>(c, s, std::make_integer_sequence<int, nv>{});
    vel_down_all<T>(c, s, std::make_integer_sequence<int, nv>{});
  } else {
    constexpr int MAXCH = max_chain<T>();
    __shared__ double s_vel[MAXCH * 6 * LBS];
    __shared__ double s_own[12 * LBS];
    if constexpr (!DIAG) {
      double E[9], r[3];
      const int jn = valid ? lane : 0;
      rbd::joint_placement(*model, jn, xg[jn] + eps, E, r);
#pragma unroll
      for (int e = 0; e < 9; ++e) s_own[lane * 12 + e] = E[e];
#pragma unroll
      for (int e = 0; e < 3; ++e) s_own[lane * 12 + 9 + e] = r[e];
    }
    CfgCtx c;
    c.m = model; c.qc0 = qc0; c.xg = xg; c.ug = ug;
    c.W = ws + (int64_t)blockIdx.x * (nv * WS_PER_JOINT * LBS) + lane;
    c.lvel = s_vel + lane;
    c.i = valid ? lane : 0; c.j = -1;
    c.own = DIAG ? nullptr : s_own + lane * 12;     // eps_mach^(1/4) steps are the ones the q-cache holds
    CfgState<T> s;
    cfg_up_all<T>(c, s, std::make_integer_sequence<int, nv>{});
#pragma unroll
    for (int e = 0; e < WS_PER_JOINT; ++e) s.w[0][e] = c.W[e * LBS];
    CfgDownOut<nv + 1> o{uq};
    cfg_down_all<T>(c, s, o, std::make_integer_sequence<int, nv>{});
  }
  __builtin_amdgcn_sched_barrier(0);
  typedef __attribute__((address_space(4))) const LinParams* kernarg_t;
  const kernarg_t kp = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  __syncthreads();
  if constexpr (!DIAG) {
    if (double* ao = kp->accel_out) {           // accelerations of the nv perturbed points of this level ([pair][3 nv directions: q, v, u | the point itself][nv]), for the analytic mode-1 pass
      ao += bt * (3 * nv + 1) * nv;
      double* al = ao + (LEVEL - 1) * nv * nv;
      for (int e = lane; e < nv * nv; e += LBS) { const int c = e / nv; al[e] = s_q[c * (nv + 1) + (e - c * nv)]; }
      // level 2: the idle lanes (no direction of theirs) have evaluated the unperturbed point into the spare row
      if (LEVEL == 2 && lane < nv) ao[3 * nv * nv + lane] = s_q[nv * (nv + 1) + lane];
      return;
    }
  }
  double* fcol = LEVEL == 3 ? kp->fu + bt * n * nv : kp->fx + bt * n * n + (LEVEL == 2 ? nv * n : 0);
  double* out = fcol;
  int64_t ostride = n;
  if constexpr (DIAG) {
    // column (d, d): f_xx at d (n + n n), f_uu at d (n + n m)
    if (LEVEL == 3) { out = kp->fuu + bt * n * nv * nv; ostride = n + (int64_t)n * nv; }
    else { ostride = n + (int64_t)n * n; out = kp->fxx + bt * n * n * n + (LEVEL == 2 ? nv * ostride : 0); }
  }
  first_output<nv, DIAG>(s_q, lane, LEVEL == 1 ? 0 : (LEVEL == 2 ? nv : n), out, ostride, fcol, kp->f_val + bt * n, xg, model->dt, eps, kp->skip_top != 0);
}

// ---- q- and v-caches ------------------------------------------------------------------------------------------------
// lin_static_qvcache_kernel: one wave per (instance, t), lane c = configuration c of the mode-2 stencil (0: q, 1+i:
// q + eps e_i).  The chain-wise sweep of the configuration level, reduced to what the caches hold: rbd::aba_qpart
// (E | r | U | 1/D | Ia per joint) and, at the base velocity, rbd::aba_vpart_cached (cb | pA0 | Ia cb) -- v-cache entry 0
// for the base configuration, nv+1+i for configuration 1+i.
template <class T, int K>
__device__ __forceinline__ void placement_static(const DevModel& m, double q, double* P) {
  const double* Rp = m.Rp[K];
  const double* a = m.axis[K];
  double* E = P;
  double* r = P + 9;
  if constexpr (!T::prismatic[K]) {
    double sn, cs;
    sincos(q, &sn, &cs);
    const double Kx[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
    double K2[9], RJ[9], Rc[9];
    rbd::mm3(Kx, Kx, K2);
    const double omc = 1.0 - cs;
#pragma unroll
    for (int k = 0; k < 9; ++k) RJ[k] = sn * Kx[k] + omc * K2[k];
    RJ[0] += 1.0; RJ[4] += 1.0; RJ[8] += 1.0;
    rbd::mm3(Rp, RJ, Rc);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) E[3 * k + l] = Rc[3 * l + k];
    r[0] = m.pp[K][0]; r[1] = m.pp[K][1]; r[2] = m.pp[K][2];
  } else {
    const double dd[3] = {a[0] * q, a[1] * q, a[2] * q};
    double Rd[3];
    rbd::mv3(Rp, dd, Rd);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) E[3 * k + l] = Rp[3 * l + k];
    r[0] = m.pp[K][0] + Rd[0]; r[1] = m.pp[K][1] + Rd[1]; r[2] = m.pp[K][2] + Rd[2];
  }
}

struct QvCtx {
  const DevModel* __restrict__ m;
  const double* __restrict__ xg;
  double* __restrict__ qc;     // this lane's q-cache block
  double* __restrict__ vc;     // this lane's v-cache block
  double* lvel;                // LDS: link velocities of the chain being swept
  int jn;                      // joint whose position this lane steps (-1: none)
  double eps;
};
template <class T>
struct QvState {
  double vel[T::N][6];
  double accI[T::N][21];
};
template <class T, int K>
__device__ __forceinline__ void qv_placement(const QvCtx& c, double* P) {
  double q = c.xg[K];
  if (K == c.jn) q = q + c.eps;
  placement_static<T, K>(*c.m, q, P);
}
template <class T, int K>
__device__ __forceinline__ void qv_joint_vel(const QvCtx& c, const double* P, const double* vel_par, double* vel) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* a = c.m->axis[K];
  const double vK = c.xg[T::N + K];
  double vJ[6] = {0, 0, 0, 0, 0, 0};
  vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
  if constexpr (T::parent[K] >= 0) rbd::xform_motion(P, P + 9, vel_par, vel);
  else {
#pragma unroll
    for (int k = 0; k < 6; ++k) vel[k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) vel[k] += vJ[k];
}
template <class T, int K>
__device__ __forceinline__ void qv_vel_from_root(const QvCtx& c, double* vel) {
  double P[12];
  qv_placement<T, K>(c, P);
  if constexpr (T::parent[K] >= 0) {
    double vp[6];
    qv_vel_from_root<T, T::parent[K]>(c, vp);
    qv_joint_vel<T, K>(c, P, vp, vel);
  } else {
    qv_joint_vel<T, K>(c, P, nullptr, vel);
  }
}
template <class T, int K, int F, int E>
__device__ __forceinline__ void qv_chain_down(const QvCtx& c, double* cur) {
  double P[12], vel[6];
  qv_placement<T, K>(c, P);
  if constexpr (T::parent[K] >= 0) qv_joint_vel<T, K>(c, P, cur, vel);
  else qv_joint_vel<T, K>(c, P, nullptr, vel);
#pragma unroll
  for (int k = 0; k < 6; ++k) { cur[k] = vel[k]; c.lvel[((K - F) * 6 + k) * LBS] = vel[k]; }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K < E) qv_chain_down<T, K + 1, F, E>(c, cur);
}
template <class T, int K, int F>
__device__ __forceinline__ void qv_chain_up(const QvCtx& c, QvState<T>& s) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* a = c.m->axis[K];
  double P[12], IA[21], U[6], Ia[21], vel[6];
  qv_placement<T, K>(c, P);
#pragma unroll
  for (int k = 0; k < 6; ++k) vel[k] = c.lvel[((K - F) * 6 + k) * LBS];
  if constexpr (has_child<T>(K)) {
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] = s.accI[K][k];
  } else {
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] = c.m->I6[K][k];
  }
  double d = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) U[r] = IA[rbd::sidx(r, o)] * a[0] + IA[rbd::sidx(r, o + 1)] * a[1] + IA[rbd::sidx(r, o + 2)] * a[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) d += a[k] * U[o + k];
  const double dinv = 1.0 / d;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int cc = 0; cc <= r; ++cc) Ia[rbd::sidx(r, cc)] = IA[rbd::sidx(r, cc)] - U[r] * U[cc] * dinv;
  constexpr int par = T::parent[K];
  if constexpr (par >= 0) {
    if constexpr (first_contrib<T>(K)) {
#pragma unroll
      for (int k = 0; k < 21; ++k) s.accI[par][k] = c.m->I6[par][k];
    }
    rbd::add_xtix(P, P + 9, Ia, s.accI[par]);
  }
  // rbd::aba_vpart_cached at the base velocity
  const double vK = c.xg[T::N + K];
  double vJ[6] = {0, 0, 0, 0, 0, 0}, cb[6], pA[6], Iv[6], Iac[6];
  vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
  rbd::crm(vel, vJ, cb);
  rbd::sym6_mv(c.m->I6[K], vel, Iv);
  rbd::crf(vel, Iv, pA);
  rbd::sym6_mv(Ia, cb, Iac);
  double* qo = c.qc + K * rbd::QC_STRIDE;
#pragma unroll
  for (int k = 0; k < 12; ++k) qo[k] = P[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) qo[12 + k] = U[k];
  qo[18] = dinv;
#pragma unroll
  for (int k = 0; k < 21; ++k) qo[19 + k] = Ia[k];
  double* vo = c.vc + K * rbd::VC_STRIDE;
#pragma unroll
  for (int k = 0; k < 6; ++k) { vo[k] = cb[k]; vo[6 + k] = pA[k]; vo[12 + k] = Iac[k]; }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (K > F) qv_chain_up<T, K - 1, F>(c, s);
}
template <class T, int K>
__device__ __forceinline__ void qv_chain_at(const QvCtx& c, QvState<T>& s) {
  if constexpr (chain_end<T>(K)) {
    constexpr int F = chain_first<T>(K), P = T::parent[F];
    double cur[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (P >= 0) {
      if constexpr (first_contrib<T>(F) && n_children<T>(P) > 1) qv_vel_from_root<T, P>(c, s.vel[P]);
#pragma unroll
      for (int k = 0; k < 6; ++k) cur[k] = s.vel[P][k];
    }
    qv_chain_down<T, F, F, K>(c, cur);
    qv_chain_up<T, K, F>(c, s);
  }
}
template <class T, int... Ks>
__device__ __forceinline__ void qv_all(const QvCtx& c, QvState<T>& s, std::integer_sequence<int, Ks...>) {
  (qv_chain_at<T, T::N - 1 - Ks>(c, s), ...);
}

template <class T>
__global__ __launch_bounds__(LBS, 2) void lin_static_qvcache_kernel(LinParams p, const DevModel* __restrict__ model, const double* __restrict__ xs,
                                                                 double* __restrict__ qcache, double* __restrict__ vcache) {
  constexpr int nv = T::N, n = 2 * nv, MAXCH = max_chain<T>();
  const int lane = threadIdx.x;
  const int64_t Tn = p.d.T;
  // a lane per (instance, t, configuration), 64 per wave: with a wave per (instance, t) 25 of its lanes (ncfg = nv + 1) or 63 of them
  // (the base configuration alone: first-order-only contexts, the accelerations of the analytic mode-1 pass) idle through a 38-joint chain
  const int64_t e = (int64_t)blockIdx.x * LBS + lane;    // evaluation: (instance, t) x configuration, 64 per wave whatever ncfg is
  const int64_t bt = e / p.ncfg;
  const int cfgi = (int)(e - bt * p.ncfg);
  const int b = (int)(bt / Tn);
  const int64_t t = bt % Tn;
  __shared__ double s_vel[MAXCH * 6 * LBS];
  if (bt >= p.d.batch * Tn) return;                       // (no workgroup barrier below)
  QvCtx c;
  c.m = model;
  c.xg = xs + ((int64_t)b * (Tn + 1) + t) * n;
  c.qc = qcache + (bt * p.ncfg + cfgi) * (int64_t)nv * rbd::QC_STRIDE;
  c.vc = vcache + (bt * p.nvcfg + (cfgi == 0 ? 0 : nv + cfgi)) * (int64_t)nv * rbd::VC_STRIDE;
  c.lvel = s_vel + lane;
  c.jn = cfgi - 1;
  c.eps = sqrt(sqrt(DBL_EPSILON));
  QvState<T> s;
  qv_all<T>(c, s, std::make_integer_sequence<int, nv>{});
}

// lin_static_vcache_kernel: v-cache entries 1 + i = (q, v + eps e_i): first pass of the ABA on the cached base q-part
template <class T, int K>
__device__ __forceinline__ void vc_joint(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ xg, int iv, double eps,
                                         double (&vel)[T::N][6], double* __restrict__ vc) {
  constexpr int o = T::prismatic[K] ? 3 : 0;
  const double* E = qc + K * rbd::QC_STRIDE;
  const double* r = E + 9;
  const double* Ia = E + 19;
  const double* a = m.axis[K];
  double vK = xg[T::N + K];
  if (K == iv) vK = vK + eps;
  double vJ[6] = {0, 0, 0, 0, 0, 0}, v[6], cb[6], pA[6], Iv[6], Iac[6];
  vJ[o] = a[0] * vK; vJ[o + 1] = a[1] * vK; vJ[o + 2] = a[2] * vK;
  if constexpr (T::parent[K] >= 0) rbd::xform_motion(E, r, vel[T::parent[K]], v);
  else {
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) { v[k] += vJ[k]; vel[K][k] = v[k]; }
  rbd::crm(v, vJ, cb);
  rbd::sym6_mv(m.I6[K], v, Iv);
  rbd::crf(v, Iv, pA);
  rbd::sym6_mv(Ia, cb, Iac);
  double* vo = vc + K * rbd::VC_STRIDE;
#pragma unroll
  for (int k = 0; k < 6; ++k) { vo[k] = cb[k]; vo[6 + k] = pA[k]; vo[12 + k] = Iac[k]; }
  __builtin_amdgcn_sched_barrier(0);
}
template <class T, int... Ks>
__device__ __forceinline__ void vc_all(const DevModel& m, const double* __restrict__ qc, const double* __restrict__ xg, int iv, double eps,
                                       double (&vel)[T::N][6], double* __restrict__ vc, std::integer_sequence<int, Ks...>) {
  (vc_joint<T, Ks>(m, qc, xg, iv, eps, vel, vc), ...);
}
template <class T>
__global__ __launch_bounds__(LBS) void lin_static_vcache_kernel(LinParams p, const DevModel* __restrict__ model, const double* __restrict__ xs,
                                                                const double* __restrict__ qcache, double* __restrict__ vcache) {
  constexpr int nv = T::N, n = 2 * nv;
  const int64_t bt = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t Tn = p.d.T;
  const int b = (int)(bt / Tn);
  const int64_t t = bt % Tn;
  if (lane >= nv) return;
  const double* __restrict__ qc = qcache + (bt * (nv + 1)) * (int64_t)nv * rbd::QC_STRIDE;
  const double* __restrict__ xg = xs + ((int64_t)b * (Tn + 1) + t) * n;
  double* __restrict__ vc = vcache + (bt * (2 * nv + 1) + 1 + lane) * (int64_t)nv * rbd::VC_STRIDE;
  double vel[nv][6];
  vc_all<T>(*model, qc, xg, lane, sqrt(sqrt(DBL_EPSILON)), vel, vc, std::make_integer_sequence<int, nv>{});
}

}  // namespace

int lin_static_supported(const DevModel& m) {
  if (m.ff) return 0;            // compiled-in topologies are trees of 1-DoF joints
  if (topo_matches<TopoTalos38>(m)) return 1;
  if (topo_matches<TopoChain6>(m)) return 2;
#define DDP_TOPO_MATCH(ID, TOPO) if (topo_matches<TOPO>(m)) return (ID) - 1;   // generated topologies: lin_path ID, internal id ID - 1
  DDP_TOPO_EXTRA(DDP_TOPO_MATCH)
#undef DDP_TOPO_MATCH
  return 0;
}

// doubles of workspace one (instance, t) needs at the configuration level
int64_t lin_static_ws_per_bt(const DevModel& m) {
  const int64_t nv = m.nv, TRI = nv * (nv - 1) / 2, GU = (TRI + LBS - 1) / LBS;
  return GU * nv * WS_PER_JOINT * LBS;
}

// level 3: torque-level points (replaces lin_offdiag_kernel<NJ, 3>)
template <class T>
static int lin_static_launch_t(ddp_hip_ctx* ctx, const LinParams& p, int level) {
  const int64_t BT = ctx->d.batch * ctx->d.T;
  constexpr int nv = T::N, TRI = nv * (nv - 1) / 2, GU = (TRI + LBS - 1) / LBS;
  if (level == 3) {
    hipLaunchKernelGGL((lin_static_tau_kernel<T, true>), dim3((unsigned)(BT * 2 * nv)), dim3(LBS), 0, ctx->stream, p);
    hipLaunchKernelGGL((lin_static_tau_kernel<T, false>), dim3((unsigned)(BT * GU)), dim3(LBS), 0, ctx->stream, p);
  } else if (level == 5) {                        // q- and v-caches
    hipLaunchKernelGGL((lin_static_qvcache_kernel<T>), dim3((unsigned)((BT * p.ncfg + LBS - 1) / LBS)), dim3(LBS), 0, ctx->stream, p, p.model, p.x, p.qcache, p.vcache);
    if (p.nvcfg > 1) hipLaunchKernelGGL((lin_static_vcache_kernel<T>), dim3((unsigned)BT), dim3(LBS), 0, ctx->stream, p, p.model, p.x, p.qcache, p.vcache);
  } else if (level == 6 || level == 7) {          // accelerations at x + sqrt(eps_mach) e_k (level 7: and u + sqrt(eps_mach) e_k) -> p.accel_out, after the base caches
    if (!p.accel_out) return DDP_HIP_E_ARG;
    hipLaunchKernelGGL((lin_static_qvcache_kernel<T>), dim3((unsigned)((BT * p.ncfg + LBS - 1) / LBS)), dim3(LBS), 0, ctx->stream, p, p.model, p.x, p.qcache, p.vcache);
    const int64_t per = ctx->lin_qws_bt * GU;
    for (int64_t bt0 = 0; bt0 < BT; bt0 += per) {
      const int64_t nb = BT - bt0 < per ? BT - bt0 : per;
      hipLaunchKernelGGL((lin_static_first_kernel<T, 1, false>), dim3((unsigned)nb), dim3(LBS), 0, ctx->stream, p, p.model, p.qcache, p.x, p.u, ctx->lin_qws, bt0);
    }
    hipLaunchKernelGGL((lin_static_first_kernel<T, 2, false>), dim3((unsigned)BT), dim3(LBS), 0, ctx->stream, p, p.model, p.qcache, p.x, p.u, ctx->lin_qws, (int64_t)0);
    if (level == 7) hipLaunchKernelGGL((lin_static_first_kernel<T, 3, false>), dim3((unsigned)BT), dim3(LBS), 0, ctx->stream, p, p.model, p.qcache, p.x, p.u, ctx->lin_qws, (int64_t)0);
  } else if (level == 0) {                        // first order
    const int64_t per = ctx->lin_qws_bt * GU;     // one wave per (instance, t) uses one of the GU workspace slots of a slice entry
    for (int64_t bt0 = 0; bt0 < BT; bt0 += per) {
      const int64_t nb = BT - bt0 < per ? BT - bt0 : per;
      hipLaunchKernelGGL((lin_static_first_kernel<T, 1, false>), dim3((unsigned)nb), dim3(LBS), 0, ctx->stream, p, p.model, p.qcache, p.x, p.u, ctx->lin_qws, bt0);
    }
    hipLaunchKernelGGL((lin_static_first_kernel<T, 2, false>), dim3((unsigned)BT), dim3(LBS), 0, ctx->stream, p, p.model, p.qcache, p.x, p.u, ctx->lin_qws, (int64_t)0);
    hipLaunchKernelGGL((lin_static_first_kernel<T, 3, false>), dim3((unsigned)BT), dim3(LBS), 0, ctx->stream, p, p.model, p.qcache, p.x, p.u, ctx->lin_qws, (int64_t)0);
  } else if (level == 4) {
    // diagonal second-order entries of the u directions; those of the q and v directions are formed by the torque-level
    // row kernel (level 3) on an otherwise idle lane, which therefore runs ahead of levels 2 and 1
    hipLaunchKernelGGL((lin_static_first_kernel<T, 3, true>), dim3((unsigned)BT), dim3(LBS), 0, ctx->stream, p, p.model, p.qcache, p.x, p.u, ctx->lin_qws, (int64_t)0);
  } else if (level == 1) {
    // In slices of (instance, t), so that the per-wave workspace stays small.  The sweep kernel of slice k+1 (one wave per
    // SIMD, long waves) and the acceleration / output kernel of slice k run on two streams with two workspaces: each
    // fills the other's tail instead of leaving the chip to drain between launches.
    const int64_t per = ctx->lin_qws_bt;
    hipStream_t s0 = ctx->stream, s1 = ctx->lin_stream2;
    int k = 0;
    for (int64_t bt0 = 0; bt0 < BT; bt0 += per, ++k) {
      const int64_t nb = BT - bt0 < per ? BT - bt0 : per;
      const int w = k & 1;
      double* ws = w ? ctx->lin_qws2 : ctx->lin_qws;
      // a failed wait / record would silently turn into a race on the shared workspace: every return code is checked
      if (k >= 2) HIP_TRY(hipStreamWaitEvent(s0, ctx->lin_ev_dn[w], 0));        // slice k-2 is done with this workspace
      hipLaunchKernelGGL((lin_static_cfg_up_kernel<T>), dim3((unsigned)(nb * GU)), dim3(LBS), 0, s0, p, p.model, p.qcache, p.x, p.u, ws, bt0);
      HIP_TRY(hipEventRecord(ctx->lin_ev_up[w], s0));
      HIP_TRY(hipStreamWaitEvent(s1, ctx->lin_ev_up[w], 0));
      hipLaunchKernelGGL((lin_static_cfg_down_kernel<T>), dim3((unsigned)(nb * GU)), dim3(LBS), 0, s1, p, p.model, p.qcache, p.x, p.u, ws, bt0);
      HIP_TRY(hipEventRecord(ctx->lin_ev_dn[w], s1));
    }
    if (k >= 1) HIP_TRY(hipStreamWaitEvent(s0, ctx->lin_ev_dn[0], 0));
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(s0, ctx->lin_ev_dn[1], 0));
  } else if (level == 2) {
    hipLaunchKernelGGL((lin_static_vel_kernel<T, true>), dim3((unsigned)(BT * nv)), dim3(LBS), 0, ctx->stream, p);
    hipLaunchKernelGGL((lin_static_vel_kernel<T, false>), dim3((unsigned)(BT * GU)), dim3(LBS), 0, ctx->stream, p);
  }
  HIP_TRY(hipGetLastError());
  return DDP_HIP_OK;
}

int lin_static_launch(ddp_hip_ctx* ctx, const LinParams& p, int level) {
  if (ctx->lin_static == 1) return lin_static_launch_t<TopoTalos38>(ctx, p, level);
  if (ctx->lin_static == 2) return lin_static_launch_t<TopoChain6>(ctx, p, level);
#define DDP_TOPO_LAUNCH(ID, TOPO) if (ctx->lin_static == (ID) - 1) return lin_static_launch_t<TOPO>(ctx, p, level);
  DDP_TOPO_EXTRA(DDP_TOPO_LAUNCH)
#undef DDP_TOPO_LAUNCH
  return DDP_HIP_E_UNSUPPORTED;
}
