// fwd.hip -- trajectory rollout, augmented cost and the forward sweep with batched line-search steps.
//
// Replaces make_trajectory (ddp.hpp:392-415), cost_seq_aug (ddp.hpp:699-735) and
// forward_pass<M> (ddp_fwd.ipp:9-67).  A rollout is sequential in t; the independent units are the
// (instance, step-size candidate) chains: one lane per chain, the candidates of one instance in
// adjacent lanes so that the gain matrices K_t are fetched once per wave.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "internal.h"
#include "rbd.h"

namespace {

struct FwdParams {
  Dims d;
  const DevModel* model;
  const int64_t* ne;
  const int64_t* Epre;
  const double* target;
  const double *x_old, *u_old;
  double *x_new, *u_new;
  const double *fb_val, *fb_jac;
  const double *mult_origin, *mult_val, *mult_jac;
  const double* mu;
  double *costs_old, *costs_new;
  double *fw_x, *fw_u, *fw_dcost;
  double* step;
  double* dcost_acc;
  int32_t* state;
  int32_t n_alpha, round;
  int32_t no_linesearch;         // ddp_fwd.ipp:61-63: the full step is taken whatever the cost does
  int32_t cost_inline;           // latency kernel: 1 = forms sum_t (cost_new - cost_old) itself (no constraints), 0 = cand_cost_kernel does
  double* fw_cost;               // [batch][n_alpha][T+1] cost terms of the candidates (constrained problems on the latency path)
};

// constraint value at solver time t: constraint_advance_time_t::eval_to (problem.hpp:563-567) applied
// eq_advance times around config_constraint_t (:792-806) or spatial_constraint_t (:679-689)
template <int NJ>
__device__ void eval_eq(const DevModel& m, const double* target, int e, const double* x, const double* u, double* out) {
  const int nx = m.nq + m.nv;
  double xa[2 * NJ + 1], xb[2 * NJ + 1];
  for (int i = 0; i < nx; ++i) xa[i] = x[i];
  for (int k = 0; k < m.eq_advance; ++k) {
    if (k + 1 < m.eq_advance) rbd::eval_f<NJ>(m, xa, u, xb);
    else { for (int i = m.nq; i < nx; ++i) xb[i] = xa[i]; rbd::eval_f_q<NJ>(m, xa, xb); }   // the constraint reads q only (rbd.h: eval_f_q)
    for (int i = 0; i < nx; ++i) xa[i] = xb[i];
  }
  if (m.eq_kind == DDP_HIP_EQ_CONFIG) {
    for (int i = 0; i < e; ++i) out[i] = xa[i] - target[i];
  } else {
    double p[3];
    rbd::frame_position<NJ>(m, xa, p, nullptr);
    for (int i = 0; i < e; ++i) out[i] = p[i] - target[i];
  }
}

// one term of cost_seq_aug (ddp.hpp:730): l + pe.ce + mu/2 |ce|^2
template <int NJ>
__device__ double stage_cost(const FwdParams& p, const DevModel& m, int b, int64_t t, const double* x, const double* u, double mu) {
  const int nv = m.nv, n = 2 * nv, nx = m.nq + nv;
  double un = 0;
  for (int i = 0; i < nv; ++i) un += u[i] * u[i];
  double cost = 0.5 * m.c * un;                                   // problem_t::l, problem.hpp:937-942
  const int e = (int)p.ne[t];
  if (e > 0) {
    double ce[NJ > 3 ? NJ : 3];
    const int64_t Eo = p.Epre[t], Etot = p.d.Etot;
    eval_eq<NJ>(m, p.target + Eo, e, x, u, ce);
    const double* org = p.mult_origin + ((int64_t)b * p.d.T + t) * nx;
    const double* val = p.mult_val + (int64_t)b * Etot + Eo;
    const double* jac = p.mult_jac + ((int64_t)b * Etot + Eo) * n;
    double dot = 0, sq = 0;
    double dxo[2 * NJ];
    if (m.ff) lie::difference_x(m, org, x, dxo);                  // x (-) origin on the group
    for (int i = 0; i < e; ++i) {
      double pe = val[i];                                         // mat_seq_common.hpp:105-115
      double s = 0;
      for (int l = 0; l < n; ++l) s += jac[i + (int64_t)l * e] * (m.ff ? dxo[l] : x[l] - org[l]);
      pe += s;
      dot += pe * ce[i];
      sq += ce[i] * ce[i];
    }
    cost += dot;
    cost += (mu / 2) * sq;
  }
  return cost;
}

template <int NJ>
__global__ void rollout_kernel(FwdParams p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.d.batch) return;
  const DevModel& m = *p.model;
  const int nx = m.nq + m.nv, nu = m.nv;
  double* xs = const_cast<double*>(p.x_old) + (int64_t)b * (p.d.T + 1) * nx;
  const double* us = p.u_old + (int64_t)b * p.d.T * nu;
  double x[2 * NJ + 1], xn[2 * NJ + 1], u[NJ];
  for (int i = 0; i < nx; ++i) x[i] = xs[i];
  for (int64_t t = 0; t < p.d.T; ++t) {
    for (int i = 0; i < nu; ++i) u[i] = us[t * nu + i];
    rbd::eval_f<NJ>(m, x, u, xn);
    for (int i = 0; i < nx; ++i) { x[i] = xn[i]; xs[(t + 1) * nx + i] = xn[i]; }
  }
}

// cost_seq_aug of one trajectory: one lane per (instance, t)
template <int NJ>
__global__ void cost_kernel(FwdParams p, int which) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  if (gid >= p.d.batch * (T + 1)) return;
  const int b = (int)(gid / (T + 1));
  const int64_t t = gid % (T + 1);
  const DevModel& m = *p.model;
  const int nx = m.nq + m.nv, nu = m.nv;
  double* out = (which == 0 ? p.costs_old : p.costs_new) + (int64_t)b * (T + 1);
  if (t == T) { out[T] = 0.0; return; }                           // problem_t::lf, problem.hpp:932-936
  const double* xs = (which == 0 ? p.x_old : p.x_new) + ((int64_t)b * (T + 1) + t) * nx;
  const double* us = (which == 0 ? p.u_old : p.u_new) + ((int64_t)b * T + t) * nu;
  double x[2 * NJ + 1], u[NJ];
  for (int i = 0; i < nx; ++i) x[i] = xs[i];
  for (int i = 0; i < nu; ++i) u[i] = us[i];
  out[t] = stage_cost<NJ>(p, m, b, t, x, u, p.mu[b]);
}

// closed-loop rollouts (ddp_fwd.ipp:39-51) of n_alpha candidate steps per instance + their summed cost
// difference (ddp_fwd.ipp:54-56)
template <int NJ>
__global__ void forward_kernel(FwdParams p) {
  // the model table in LDS: the dynamics of every step read it joint by joint (axis, placement, inertia: some 40 words per joint
  // and evaluation), and from global memory each of those reads is a dependent L2 round trip of the one lane that rolls out
  __shared__ DevModel s_model;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.model);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s_model);
    for (unsigned i = threadIdx.x; i < sizeof(DevModel) / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int na = p.n_alpha;
  if (gid >= p.d.batch * na) return;
  const int b = gid / na, a = gid % na;
  if (p.state[b] != 0) return;
  const int cand = p.round * na + a;
  if (cand > 33) { p.fw_dcost[(int64_t)b * na + a] = INFINITY; return; }   // 2^-34 < 1e-10: never tried (ddp_fwd.ipp:35-37)
  const double step = ldexp(1.0, -cand);
  const DevModel& m = s_model;
  const int nv = m.nv, n = 2 * nv, nx = m.nq + nv, nu = nv;
  const int64_t T = p.d.T;
  const double mu = p.mu[b];
  const double* xo = p.x_old + (int64_t)b * (T + 1) * nx;
  const double* uo = p.u_old + (int64_t)b * T * nu;
  double* xw = p.fw_x + ((int64_t)b * na + a) * (T + 1) * nx;
  double* uw = p.fw_u + ((int64_t)b * na + a) * T * nu;
  const double* cold = p.costs_old + (int64_t)b * (T + 1);
  double x[2 * NJ + 1], xn[2 * NJ + 1], u[NJ], dx[2 * NJ];
  const double* x0 = p.x_new + (int64_t)b * (T + 1) * nx;        // x_new,0 is preset by the caller (ddp.hpp:752)
  for (int i = 0; i < nx; ++i) { x[i] = x0[i]; xw[i] = x0[i]; }
  double dsum = 0.0;
  for (int64_t t = 0; t < T; ++t) {
    const double* k = p.fb_val + ((int64_t)b * T + t) * nu;
    const double* K = p.fb_jac + ((int64_t)b * T + t) * nu * n;
    if (m.ff) lie::difference_x(m, xo + t * nx, x, dx);                       // :45 difference(out, old, new)
    else for (int i = 0; i < n; ++i) dx[i] = x[i] - xo[t * nx + i];
    for (int i = 0; i < nu; ++i) u[i] = uo[t * nu + i] + step * k[i];         // :47-48
    for (int i = 0; i < nu; ++i) {
      double s = 0;
      for (int l = 0; l < n; ++l) s += K[i + l * nu] * dx[l];
      u[i] += s;                                                              // :49
    }
    for (int i = 0; i < nu; ++i) uw[t * nu + i] = u[i];
    const double c_new = stage_cost<NJ>(p, m, b, t, x, u, mu);
    dsum += c_new - cold[t];
    rbd::eval_f<NJ>(m, x, u, xn);                                             // :50
    for (int i = 0; i < nx; ++i) { x[i] = xn[i]; xw[(t + 1) * nx + i] = xn[i]; }
  }
  dsum += 0.0 - cold[T];
  p.fw_dcost[(int64_t)b * na + a] = dsum;
}

// Latency path of the same rollouts (tree models, no constraints): one 64-lane workgroup per instance, 8 lanes per
// candidate.  The gain product K_t (x_new - x_old) is spread over the 8 lanes of a candidate (each takes every 8th
// column of K_t, contiguous column loads, then three xor-shuffles); the forward dynamics keeps its per-joint state in
// LDS and its 8 lanes walk the tree level by level (rbd::aba_tree_coop), so the legs, arms and head advance together.
template <int NJ>
__global__ __launch_bounds__(64) void forward_kernel_lat(FwdParams p) {
  constexpr int NC = 8, NH = 8;                      // candidates per instance, helper lanes per candidate
  constexpr int n = 2 * NJ, nx = 2 * NJ, nu = NJ;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* aba_state = lds;                           // ABA_LDS_SLOTS * NJ * NC
  double* s_dx = lds + rbd::ABA_LDS_SLOTS * NJ * NC; // NC * n
  double* s_x = s_dx + NC * n;                       // NC * nx
  double* s_u = s_x + NC * nx;                       // NC * nu
  double* s_qdd = s_u + NC * nu;                     // NC * nu
  const int b = blockIdx.x;
  if (p.state[b] != 0) return;
  const int tid = threadIdx.x, a = tid / NH, h = tid % NH;
  const int na = p.n_alpha;
  const int cand = p.round * na + a;
  const bool live = a < na && cand <= 33;            // 2^-34 < 1e-10: never tried (ddp_fwd.ipp:35-37)
  if (a < na && cand > 33 && h == 0) p.fw_dcost[(int64_t)b * na + a] = INFINITY;
  const double step = ldexp(1.0, -cand);
  const DevModel& m = *p.model;
  const int64_t T = p.d.T;
  const double* xo = p.x_old + (int64_t)b * (T + 1) * nx;
  const double* uo = p.u_old + (int64_t)b * T * nu;
  double* xw = p.fw_x + ((int64_t)b * na + (a < na ? a : 0)) * (T + 1) * nx;
  double* uw = p.fw_u + ((int64_t)b * na + (a < na ? a : 0)) * T * nu;
  const double* cold = p.costs_old + (int64_t)b * (T + 1);
  double* dx = s_dx + a * n;
  double* x = s_x + a * nx;
  double* u = s_u + a * nu;
  double* qdd = s_qdd + a * nu;
  const double* x0 = p.x_new + (int64_t)b * (T + 1) * nx;        // x_new,0 is preset by the caller (ddp.hpp:752)
  if (live)
    for (int i = h; i < nx; i += NH) { x[i] = x0[i]; xw[i] = x0[i]; }
  double dsum = 0.0;
  __syncthreads();
  for (int64_t t = 0; t < T; ++t) {
    const double* k = p.fb_val + ((int64_t)b * T + t) * nu;
    const double* K = p.fb_jac + ((int64_t)b * T + t) * nu * n;
    if (live)
      for (int i = h; i < n; i += NH) dx[i] = x[i] - xo[t * nx + i];           // :45 difference(out, old, new)
    __syncthreads();
    double acc[NJ];
#pragma unroll
    for (int i = 0; i < nu; ++i) acc[i] = 0.0;
    for (int l = h; l < n; l += NH) {
      const double d = dx[l];
      const double* Kc = K + (int64_t)l * nu;
#pragma unroll
      for (int i = 0; i < nu; ++i) acc[i] += Kc[i] * d;
    }
#pragma unroll
    for (int i = 0; i < nu; ++i) {
      acc[i] += __shfl_xor(acc[i], 1, 64);
      acc[i] += __shfl_xor(acc[i], 2, 64);
      acc[i] += __shfl_xor(acc[i], 4, 64);
    }
    if (h == 0 && live) {
      double un = 0;
#pragma unroll
      for (int i = 0; i < nu; ++i) {
        double ui = uo[t * nu + i] + step * k[i];                               // :47-48
        ui += acc[i];                                                           // :49
        u[i] = ui;
        uw[t * nu + i] = ui;
        un += ui * ui;
      }
      const double c_new = 0.5 * m.c * un;                                      // problem_t::l (no constraints on this path)
      dsum += c_new - cold[t];
    }
    __syncthreads();
    rbd::aba_tree_coop<NJ, NC, NH>(m, x, x + NJ, u, qdd, aba_state, a, h, live);   // :50 (ends with a barrier)
    if (live)
      for (int i = h; i < NJ; i += NH) {                                        // dynamics_t::eval_to, problem.hpp:441-461
        const double vo = m.dt * x[NJ + i];
        const double qn = x[i] + vo;
        const double vn = x[NJ + i] + qdd[i] * m.dt;
        x[i] = qn; x[NJ + i] = vn;
        xw[(t + 1) * nx + i] = qn; xw[(t + 1) * nx + NJ + i] = vn;
      }
    __syncthreads();
  }
  if (h == 0 && live) {
    dsum += 0.0 - cold[T];
    p.fw_dcost[(int64_t)b * na + a] = dsum;
  }
}

// Second latency path: one 64-lane workgroup (= one wave) per (instance, four candidates), 16 lanes per candidate.
// What the first one waits for at every step is global memory: the 23 KB gain matrix K_t (written by the backward sweep a whole
// linearisation ago: an HBM read in the middle of the step), k_t, u_old, x_old, and the per-level reads of the model tables
// inside the traversal (dependent L2 round trips, three per tree level).  Here
//   * the model (inertias, axes, placements, level tables: rbd::CoopModel) is copied to LDS once per launch;
//   * K_{t+1}, k_{t+1}, u_old,t+1, x_old,t+1 are requested right after the control update of step t, travel while the forward
//     dynamics of step t run, and are parked in LDS at the end of the step (registers are the second buffer);
//   * the workgroup is a single wave, so the exchange points of the traversal are compiler fences, not s_barrier + vmcnt(0)
//     (rbd::coop_sync): the prefetch stays in flight through them;
//   * the gain product is row-parallel: lane h of a candidate owns rows h, h + 16, h + 32 of K_t dx and runs down the columns
//     in order (LDS reads, conflict-free; the four candidates read the same words).
// Four candidates per workgroup halve the per-joint state (72 KB), which is what makes room for K_t and the model.
template <int NJ>
struct FwdLat2Lds {
  static constexpr int NC = 4, NH = 16, n = 2 * NJ + 1, nu = NJ;   // n: room for the state of a free-flyer model (nq = nv + 1)
  double state[rbd::ABA_LDS_SLOTS2 * (NJ + NJ / 8) * NC];      // (+ NJ / 8: the bank skew of rbd::aba_tree_coop2w)
  double K[nu * n];
  double k[nu], uo[nu], xo[n];
  double dx[NC * n], x[NC * n], u[NC * nu], qdd[NC * nu];
  rbd::CoopModel<NJ> model;
};

#ifdef FWD_STAMPS
__device__ unsigned long long g_fwd_stamps[12];
#endif

// OPEN: the open-loop rollout of make_trajectory (ddp.hpp:392-415) on the same machinery: one candidate, u = U as given, x to X
// FF: free-flyer root (nq = nv + 1): x_new (-) x_old and q (+) dt v go through SE(3) for the root (lie.h), the dynamics through
// rbd::aba_tree_coop2w's free-flyer form
template <int NJ, bool OPEN = false, bool FF = false>
__global__ __launch_bounds__(128) void forward_kernel_lat2(FwdParams p) {
  using L = FwdLat2Lds<NJ>;
  constexpr int NC = L::NC, NH = L::NH;
  constexpr int n = 2 * NJ, nq = FF ? NJ + 1 : NJ, nx = nq + NJ, nu = NJ, XS = L::n;   // XS: stride of a candidate's state in LDS
  constexpr int K2 = nu * n / 2, KR = (K2 + 63) / 64;        // K_t as 16-byte words; words per lane
  static_assert((nu * n) % 2 == 0, "K_t is moved in 16-byte words");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  L& S = *reinterpret_cast<L*>(lds);
  const int b = OPEN ? blockIdx.x : blockIdx.x / 2, half = OPEN ? 0 : blockIdx.x % 2;
  if (!OPEN && p.state[b] != 0) return;
  const int na = OPEN ? 1 : p.n_alpha;
  if (half * NC >= na) return;
  // two waves: wave 0 runs the rollout, wave 1 joins it for the inertia half of the leaf -> root pass (rbd::aba_tree_coop2w)
  const int wave = threadIdx.x / 64, tid = threadIdx.x % 64, al = tid / NH, h = tid % NH;
  const int a = half * NC + al;
  const int cand = p.round * na + a;
  const bool live = a < na && cand <= 33;            // 2^-34 < 1e-10: never tried (ddp_fwd.ipp:35-37)
  if (!OPEN && a < na && cand > 33 && h == 0 && wave == 0) p.fw_dcost[(int64_t)b * na + a] = INFINITY;
  const double step = ldexp(1.0, -cand);
  const int64_t T = p.d.T;
  const double* xo = p.x_old + (int64_t)b * (T + 1) * nx;
  const double* uo = p.u_old + (int64_t)b * T * nu;
  double* xw = OPEN ? const_cast<double*>(p.x_old) + (int64_t)b * (T + 1) * nx : p.fw_x + ((int64_t)b * na + (a < na ? a : 0)) * (T + 1) * nx;
  double* uw = OPEN ? nullptr : p.fw_u + ((int64_t)b * na + (a < na ? a : 0)) * T * nu;
  const double* cold = p.costs_old + (int64_t)b * (T + 1);
  const double* kg = p.fb_val + (int64_t)b * T * nu;
  const double* Kg = p.fb_jac + (int64_t)b * T * nu * n;
  double* dx = S.dx + al * XS;
  double* x = S.x + al * XS;
  double* u = S.u + al * nu;
  double* qdd = S.qdd + al * nu;
  if (wave == 0) {
    const DevModel& m = *p.model;
    rbd::CoopModel<NJ>& cm = S.model;
    for (int i = tid; i < NJ; i += 64) {
      for (int k2 = 0; k2 < 21; ++k2) cm.I6[i][k2] = m.I6[i][k2];
      for (int k2 = 0; k2 < 9; ++k2) cm.Rp[i][k2] = m.Rp[i][k2];
      for (int k2 = 0; k2 < 3; ++k2) { cm.axis[i][k2] = m.axis[i][k2]; cm.pp[i][k2] = m.pp[i][k2]; }
      cm.parent[i] = m.parent[i]; cm.jtype[i] = m.jtype[i];
      cm.lvl_joint[i] = m.lvl_joint[i]; cm.child_list[i] = m.child_list[i];
      cm.lvl_start[i] = m.lvl_start[i]; cm.child_start[i] = m.child_start[i];
    }
    if (tid == 0) {
      cm.lvl_start[NJ] = m.lvl_start[NJ]; cm.child_start[NJ] = m.child_start[NJ];
      cm.n_levels = m.n_levels; cm.nv = m.nv; cm.nj = m.nj;
      cm.gravity[0] = m.gravity[0]; cm.gravity[1] = m.gravity[1]; cm.gravity[2] = m.gravity[2];
      cm.dt = m.dt; cm.c = m.c;
    }
    // the role words (rbd::coop_role): lane (L, hh) of the wave takes level L's hh-th joint straight from the global tables
    for (int e = tid; e < 16 * NH; e += 64) {
      const int L = e / NH, hh = e % NH;
      unsigned long long r = 255;
      if (L < m.n_levels && m.lvl_start[L] + hh < m.lvl_start[L + 1]) {
        const int j = m.lvl_joint[m.lvl_start[L] + hh];
        int ch[rbd::ROLE_MAX_CHILDREN] = {0, 0, 0};
        const int nch = m.child_start[j + 1] - m.child_start[j];
        for (int c = 0; c < nch && c < rbd::ROLE_MAX_CHILDREN; ++c) ch[c] = m.child_list[m.child_start[j] + c];
        r = rbd::coop_role(j, m.parent[j], m.jtype[j] == DDP_HIP_JOINT_REVOLUTE, nch, ch);
      }
      cm.role[e] = r;
    }
  }
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 Kreg[KR];
  double kreg = 0.0, uoreg = 0.0, xoreg0 = 0.0, xoreg1 = 0.0, coldreg = 0.0;
  auto request = [&](int64_t t) {                    // step t's operands: K_t, k_t, u_old,t, x_old,t, the old cost term
    const int iu = tid < nu ? tid : nu - 1;
    uoreg = uo[t * nu + iu];
    if constexpr (!OPEN) {
      const d2* Kt = reinterpret_cast<const d2*>(Kg + t * nu * n);
#pragma unroll
      for (int j = 0; j < KR; ++j) { const int e = j * 64 + tid; Kreg[j] = Kt[e < K2 ? e : K2 - 1]; }
      kreg = kg[t * nu + iu];
      xoreg0 = xo[t * nx + (tid < nx ? tid : nx - 1)];
      xoreg1 = xo[t * nx + (64 + tid < nx ? 64 + tid : nx - 1)];
      coldreg = cold[t];
    }
  };
  auto park = [&]() {
    if (tid < nu) S.uo[tid] = uoreg;
    if constexpr (!OPEN) {
      d2* Ks = reinterpret_cast<d2*>(S.K);
#pragma unroll
      for (int j = 0; j < KR; ++j) { const int e = j * 64 + tid; if (e < K2) Ks[e] = Kreg[j]; }
      if (tid < nu) S.k[tid] = kreg;
      if (tid < nx) S.xo[tid] = xoreg0;
      if (64 + tid < nx) S.xo[64 + tid] = xoreg1;
    }
  };
  static_assert(nx <= 128, "x_old is parked by two words per lane");
  double cold_t = 0.0, dsum = 0.0;
  const double mc = p.model->c, mdt = p.model->dt;
  if (wave == 0) {
    request(0);
    const double* x0 = OPEN ? xw : p.x_new + (int64_t)b * (T + 1) * nx;   // x_new,0 is preset by the caller (ddp.hpp:752)
    if (live)
      for (int i = h; i < nx; i += NH) { const double v = x0[i]; x[i] = v; if (!OPEN) xw[i] = v; }
    park();
    cold_t = coldreg;
  }
  rbd::wg_sync_lds();                              // the model tables are in LDS for both waves
  // one loop, one call site of the traversal for all waves (a second call site keeps the compiler from inlining it): the helper
  // waves skip the rollout's own parts and meet wave 0 at the traversal's workgroup barriers
  const bool lead = wave == 0;
  rbd::FwdStamp* fs = nullptr;
#ifdef FWD_STAMPS
  rbd::FwdStamp fsv{};
  if (lead) fs = &fsv;
  fsv.last = wall_clock64();
#endif
  for (int64_t t = 0; t < T; ++t) {
    if (lead) {
    if constexpr (OPEN) {
      if (live)
        for (int i = h; i < nu; i += NH) u[i] = S.uo[i];
    } else {
    if constexpr (FF) {
      if (live) {                                                              // :45 difference(out, old, new) on SE(3) x R^(nv-6) x R^nv
        if (h == 0) lie::se3_difference(S.xo, x, dx);
        for (int i = 6 + h; i < NJ; i += NH) dx[i] = x[i + 1] - S.xo[i + 1];
        for (int i = h; i < NJ; i += NH) dx[NJ + i] = x[nq + i] - S.xo[nq + i];
      }
    } else {
    if (live)
      for (int i = h; i < n; i += NH) dx[i] = x[i] - S.xo[i];                  // :45 difference(out, old, new)
    }
    rbd::coop_sync<true>();
    {
      constexpr int NR = (nu + NH - 1) / NH;
      double acc[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] = 0.0;
      // fully unrolled: every LDS address is the lane's base plus an immediate (rolled, the loop spent 60 instructions per
      // column on address arithmetic for 3 multiply-adds)
      const double* Kr[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) { const int i = h + NH * r; Kr[r] = S.K + (i < nu ? i : nu - 1); }
#pragma unroll
      for (int l = 0; l < n; ++l) {
        const double d = dx[l];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] += Kr[r][l * nu] * d;
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = h + NH * r;
        if (i < nu && live) {
          double ui = S.uo[i] + step * S.k[i];                                  // :47-48
          ui += acc[r];                                                         // :49
          u[i] = ui;
          uw[t * nu + i] = ui;
        }
      }
    }
    rbd::coop_sync<true>();
    FSTAMP(fs, 0);
    if (h == 0 && live && p.cost_inline) {
      double un = 0;
      for (int i = 0; i < nu; ++i) un += u[i] * u[i];
      const double c_new = 0.5 * mc * un;                                       // problem_t::l (constrained problems: cand_cost_kernel)
      dsum += c_new - cold_t;
    }
    }
    rbd::coop_sync<true>();                          // K_t, k_t, ... have been read: their places are free for step t + 1
    FSTAMP(fs, 1);
    if (t + 1 < T) request(t + 1);
    FSTAMP(fs, 2);
    }
    rbd::aba_tree_coop2w<NJ, NC, NH, rbd::CoopModel<NJ>, FF>(S.model, x, x + nq, u, qdd, S.state, al, h, live, wave, fs);   // :50
    if (!lead) continue;
    if constexpr (FF) {
      // dynamics_t::eval_to on the group (problem.hpp:441-461, rbd::eval_f's free-flyer branch): the root's pose by one lane,
      // ahead of the velocity updates it reads
      double q7[7];
      if (live && h == 0) {
        double dq[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) dq[k] = mdt * x[nq + k];
        lie::se3_integrate(x, dq, q7);
      }
      rbd::coop_sync<true>();
      if (live) {
        for (int i = 6 + h; i < NJ; i += NH) { const double vo = mdt * x[nq + i]; const double qn = x[i + 1] + vo; x[i + 1] = qn; xw[(t + 1) * nx + i + 1] = qn; }
        for (int i = h; i < NJ; i += NH) { const double vn = x[nq + i] + qdd[i] * mdt; x[nq + i] = vn; xw[(t + 1) * nx + nq + i] = vn; }
        if (h == 0) {
#pragma unroll
          for (int k = 0; k < 7; ++k) { x[k] = q7[k]; xw[(t + 1) * nx + k] = q7[k]; }
        }
      }
    } else {
    if (live)
      for (int i = h; i < NJ; i += NH) {                                        // dynamics_t::eval_to, problem.hpp:441-461
        const double vo = mdt * x[NJ + i];
        const double qn = x[i] + vo;
        const double vn = x[NJ + i] + qdd[i] * mdt;
        x[i] = qn; x[NJ + i] = vn;
        xw[(t + 1) * nx + i] = qn; xw[(t + 1) * nx + NJ + i] = vn;
      }
    }
    FSTAMP(fs, 7);
    if (t + 1 < T) { park(); cold_t = coldreg; }
    rbd::coop_sync<true>();
    FSTAMP(fs, 8);
  }
#ifdef FWD_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0)
    for (int i = 0; i < 12; ++i) g_fwd_stamps[i] = fsv.acc[i];
#endif
  if (!OPEN && h == 0 && live && lead && p.cost_inline) {
    dsum += 0.0 - cold[T];
    p.fw_dcost[(int64_t)b * na + a] = dsum;
  }
}

// Constrained problems on the latency path.  Only the rollout is sequential in t; the cost terms of a rolled-out candidate
// (cost_seq_aug, ddp.hpp:699-735: l + pe . ce + mu/2 |ce|^2, with ce_t = eq(t, x_t, u_t) two look-ahead dynamics steps away,
// problem.hpp:563-567) are independent across t: one lane per (instance, candidate, t) ...
template <int NJ>
__global__ void cand_cost_kernel(FwdParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const int na = p.n_alpha;
  if (gid >= p.d.batch * na * T) return;
  const int64_t t = gid % T;
  const int a = (int)((gid / T) % na);
  const int b = (int)(gid / (T * na));
  if (p.state[b] != 0 || p.round * na + a > 33) return;
  const DevModel& m = *p.model;
  const int nx = m.nq + m.nv, nu = m.nv;
  const double* xs = p.fw_x + (((int64_t)b * na + a) * (T + 1) + t) * nx;
  const double* us = p.fw_u + (((int64_t)b * na + a) * T + t) * nu;
  double x[2 * NJ + 1], u[NJ];
  for (int i = 0; i < nx; ++i) x[i] = xs[i];
  for (int i = 0; i < nu; ++i) u[i] = us[i];
  p.fw_cost[((int64_t)b * na + a) * (T + 1) + t] = stage_cost<NJ>(p, m, b, t, x, u, p.mu[b]);
}
// ... and one lane per (instance, candidate) adds the differences up in the order of forward_kernel (ddp_fwd.ipp:54-56)
__global__ void cand_sum_kernel(FwdParams p) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int na = p.n_alpha;
  if (gid >= p.d.batch * na) return;
  const int b = gid / na, a = gid % na;
  if (p.state[b] != 0 || p.round * na + a > 33) return;
  const int64_t T = p.d.T;
  const double* cold = p.costs_old + (int64_t)b * (T + 1);
  const double* cnew = p.fw_cost + ((int64_t)b * na + a) * (T + 1);
  double dsum = 0.0;
  for (int64_t t = 0; t < T; ++t) dsum += cnew[t] - cold[t];
  dsum += 0.0 - cold[T];
  p.fw_dcost[(int64_t)b * na + a] = dsum;
}

// accept rule (ddp_fwd.ipp:56-60): the first (= largest) candidate with sum(new - old) <= 0; the winner's
// trajectory becomes (X_NEW, U_NEW).  grid = batch.
__global__ void select_kernel(FwdParams p) {
  const int b = blockIdx.x;
  // every wave must see the state as it was at launch: thread 0 rewrites it below, and a wave scheduled late would
  // otherwise leave before its share of the trajectory copy
  __shared__ int s_state, s_win, s_last;
  if (threadIdx.x == 0) s_state = p.state[b];
  __syncthreads();
  if (s_state != 0) return;
  const int na = p.n_alpha;
  const int64_t T = p.d.T;
  const int nx = (int)p.d.nx, nu = (int)p.d.m;
  if (threadIdx.x == 0) {
    int win = -1, last = -1;
    for (int a = 0; a < na; ++a) {
      const int cand = p.round * na + a;
      if (cand > 33) break;
      last = a;
      if (p.no_linesearch || p.fw_dcost[(int64_t)b * na + a] <= 0) { win = a; break; }
    }
    s_win = win;
    s_last = last;
    const bool floor_hit = (p.round * na + na - 1) >= 33;
    if (win >= 0) {
      p.state[b] = 1;
      p.step[b] = ldexp(1.0, -(p.round * na + win));
      p.dcost_acc[b] = p.fw_dcost[(int64_t)b * na + win];
    } else if (floor_hit) {
      p.state[b] = 2;
      p.step[b] = ldexp(1.0, -34);                 // the value `step` holds when the loop gives up
      p.dcost_acc[b] = last >= 0 ? p.fw_dcost[(int64_t)b * na + last] : 0.0;
    }
  }
  __syncthreads();
  const int src = s_win >= 0 ? s_win : s_last;     // on failure new_traj holds the last rollout tried
  if (src < 0) return;
  if (s_win < 0 && (p.round * na + na - 1) < 33) return;
  const double* xw = p.fw_x + ((int64_t)b * na + src) * (T + 1) * nx;
  const double* uw = p.fw_u + ((int64_t)b * na + src) * T * nu;
  double* xn = p.x_new + (int64_t)b * (T + 1) * nx;
  double* un = p.u_new + (int64_t)b * T * nu;
  for (int64_t i = threadIdx.x; i < (T + 1) * nx; i += blockDim.x) xn[i] = xw[i];
  for (int64_t i = threadIdx.x; i < T * nu; i += blockDim.x) un[i] = uw[i];
}

FwdParams make_params(ddp_hip_ctx* ctx) {
  FwdParams p{};
  p.d = ctx->d;
  p.model = ctx->model_d;
  p.ne = ctx->ne_d;
  p.Epre = ctx->Epre_d;
  p.target = ctx->target_d;
  auto S = [&](int s) { return ctx->seq[s].ptr; };
  p.x_old = S(DDP_HIP_SEQ_X); p.u_old = S(DDP_HIP_SEQ_U);
  p.x_new = S(DDP_HIP_SEQ_X_NEW); p.u_new = S(DDP_HIP_SEQ_U_NEW);
  p.fb_val = S(DDP_HIP_SEQ_FB_VAL); p.fb_jac = S(DDP_HIP_SEQ_FB_JAC);
  p.mult_origin = S(DDP_HIP_SEQ_MULT_ORIGIN); p.mult_val = S(DDP_HIP_SEQ_MULT_VAL); p.mult_jac = S(DDP_HIP_SEQ_MULT_JAC);
  p.mu = ctx->mu_d;
  p.costs_old = S(DDP_HIP_SEQ_COSTS_OLD); p.costs_new = S(DDP_HIP_SEQ_COSTS_NEW);
  p.fw_x = ctx->fw_x; p.fw_u = ctx->fw_u; p.fw_dcost = ctx->fw_dcost;
  p.step = ctx->step_d; p.dcost_acc = ctx->fw_dcost_acc_d; p.state = ctx->fw_state_d;
  p.n_alpha = ctx->n_alpha_max;
  p.cost_inline = ctx->d.Etot == 0 ? 1 : 0;
  p.fw_cost = ctx->fw_cost;
  p.round = 0;
  return p;
}

#define DISPATCH_NJ(nv, CALL)                 \
  do {                                        \
    if ((nv) <= 1) { CALL(1); }               \
    else if ((nv) <= 6) { CALL(6); }          \
    else if ((nv) <= 38) { CALL(38); }        \
    else { CALL(64); }                        \
  } while (0)

}  // namespace

bool fwd_lat_supported(const ddp_hip_ctx* ctx) {
  const DevModel& m = ctx->model_h;
  // (constrained problems: the rollout runs on the latency kernel, the candidates' cost terms on cand_cost_kernel)
  if (m.kind != DDP_HIP_MODEL_TREE || ctx->d.nv != 38 || (m.ff && getenv("DDP_HIP_FWD_FF_SCRATCH") != nullptr) || getenv("DDP_HIP_FWD_SCRATCH") != nullptr) return false;
  if (ctx->d.Etot != 0 && getenv("DDP_HIP_FWD_EQ_SCRATCH") != nullptr) return false;   // development: round 2's one-lane-per-rollout kernel for constrained problems
  // the cooperative traversal: at most 8 joints per tree level (one helper lane each), 16 levels and 3 children per joint
  // (rbd::coop_role packs a lane's joint of a level into one word)
  if (m.max_level_width > 8 || m.n_levels > 16) return false;
  for (int j = 0; j < m.nj; ++j)
    if (m.child_start[j + 1] - m.child_start[j] > rbd::ROLE_MAX_CHILDREN) return false;
  return true;
}

int fwd_setup(ddp_hip_ctx* ctx) {
  const Dims& d = ctx->d;
  const int64_t B = d.batch, na = ctx->n_alpha_max;
  HIP_TRY(hipMalloc(&ctx->fw_x, sizeof(double) * (size_t)(B * na * (d.T + 1) * d.nx)));
  HIP_TRY(hipMalloc(&ctx->fw_u, sizeof(double) * (size_t)(B * na * d.T * d.m)));
  HIP_TRY(hipMalloc(&ctx->fw_dcost, sizeof(double) * (size_t)(B * na)));
  if (d.Etot > 0) HIP_TRY(hipMalloc(&ctx->fw_cost, sizeof(double) * (size_t)(B * na * (d.T + 1))));
  HIP_TRY(hipMalloc(&ctx->step_d, sizeof(double) * (size_t)B));
  HIP_TRY(hipMalloc(&ctx->fw_dcost_acc_d, sizeof(double) * (size_t)B));
  HIP_TRY(hipMalloc(&ctx->fw_state_d, sizeof(int32_t) * (size_t)B));
  if (d.nv == 38) {
    // per device, by every context (the attribute is not process-wide)
    const size_t lds = sizeof(double) * (size_t)(rbd::ABA_LDS_SLOTS * 38 * 8 + 8 * (76 + 76 + 38 + 38));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel_lat<38>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel_lat2<38>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(FwdLat2Lds<38>)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel_lat2<38, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(FwdLat2Lds<38>)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel_lat2<38, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(FwdLat2Lds<38>)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&forward_kernel_lat2<38, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(FwdLat2Lds<38>)));
  }
  return DDP_HIP_OK;
}

void fwd_teardown(ddp_hip_ctx* ctx) {
  if (ctx->fw_x) (void)hipFree(ctx->fw_x);
  if (ctx->fw_u) (void)hipFree(ctx->fw_u);
  if (ctx->fw_dcost) (void)hipFree(ctx->fw_dcost);
  if (ctx->fw_cost) (void)hipFree(ctx->fw_cost);
  if (ctx->step_d) (void)hipFree(ctx->step_d);
  if (ctx->fw_dcost_acc_d) (void)hipFree(ctx->fw_dcost_acc_d);
  if (ctx->pick_pair_d) (void)hipFree(ctx->pick_pair_d);
  if (ctx->fw_state_d) (void)hipFree(ctx->fw_state_d);
}

#ifdef FWD_STAMPS
extern "C" int ddp_hip_debug_fwd_stamps(unsigned long long* out12) {
  return hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_fwd_stamps), sizeof(unsigned long long) * 12) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int ddp_hip_rollout(ddp_hip_ctx* ctx) {
  if (!ctx) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  FwdParams p = make_params(ctx);
  if (fwd_lat_supported(ctx) && getenv("DDP_HIP_FWD_LAT1") == nullptr) {
    // unconstrained trees of the Talos size: the open-loop form of the latency kernel (one workgroup per instance)
    if (ctx->model_h.ff) hipLaunchKernelGGL((forward_kernel_lat2<38, true, true>), dim3((unsigned)ctx->d.batch), dim3(128), sizeof(FwdLat2Lds<38>), ctx->stream, p);
    else hipLaunchKernelGGL((forward_kernel_lat2<38, true>), dim3((unsigned)ctx->d.batch), dim3(128), sizeof(FwdLat2Lds<38>), ctx->stream, p);
  } else {
    const int bs = 64;
    const unsigned grid = (unsigned)((ctx->d.batch + bs - 1) / bs);
#define CALL(NJ) hipLaunchKernelGGL((rollout_kernel<NJ>), dim3(grid), dim3(bs), 0, ctx->stream, p)
    DISPATCH_NJ(ctx->d.nv, CALL);
#undef CALL
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DDP_HIP_OK;
}

static int launch_cost(ddp_hip_ctx* ctx, FwdParams& p, int which) {
  const int bs = 64;
  const int64_t total = ctx->d.batch * (ctx->d.T + 1);
  const unsigned grid = (unsigned)((total + bs - 1) / bs);
#define CALL(NJ) hipLaunchKernelGGL((cost_kernel<NJ>), dim3(grid), dim3(bs), 0, ctx->stream, p, which)
  DISPATCH_NJ(ctx->d.nv, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_cost_seq_aug(ddp_hip_ctx* ctx, int which, const double* mu) {
  if (!ctx || !mu || (which != 0 && which != 1)) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(ctx->mu_d, mu, sizeof(double) * (size_t)ctx->d.batch, hipMemcpyHostToDevice, ctx->stream));
  FwdParams p = make_params(ctx);
  int rc = launch_cost(ctx, p, which);
  if (rc != DDP_HIP_OK) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_forward(ddp_hip_ctx* ctx, const double* mu, int32_t n_alpha, double* step_out, double* dcost_out) {
  if (!ctx || !mu || !step_out || n_alpha < 0 || n_alpha > ctx->n_alpha_max) return DDP_HIP_E_ARG;
  // n_alpha == 0: do_linesearch == false (ddp_fwd.ipp:61-63) -- one rollout at step 1, accepted unconditionally
  const bool no_linesearch = n_alpha == 0;
  if (no_linesearch) n_alpha = 1;
  const Dims& d = ctx->d;
  const int64_t B = d.batch;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(ctx->mu_d, mu, sizeof(double) * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
  std::vector<int32_t> state((size_t)B);
  for (int64_t b = 0; b < B; ++b) state[(size_t)b] = ctx->active_h[(size_t)b] ? 0 : 1;   // a frozen instance is not searched
  HIP_TRY(hipMemcpyAsync(ctx->fw_state_d, state.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
  FwdParams p = make_params(ctx);
  p.n_alpha = n_alpha;
  p.no_linesearch = no_linesearch ? 1 : 0;
  int rc = launch_cost(ctx, p, 0);                                   // ddp_fwd.ipp:24-26
  if (rc != DDP_HIP_OK) return rc;
  const int bs = 64;
  const unsigned grid = (unsigned)((B * n_alpha + bs - 1) / bs);
  bool floor_hit = false;
  for (int round = 0; round * n_alpha <= 33; ++round) {
    p.round = round;
    prof_begin(ctx, DDP_HIP_K_FWD_ROLLOUT);
    // tree models of the Talos size: the latency path (two workgroups per instance, 16 lanes per candidate); with constraints the
    // cost terms of the rolled-out candidates come from cand_cost_kernel (parallel over t) instead of the rollout itself
    const bool lat_path = fwd_lat_supported(ctx) && n_alpha <= 8;
    if (lat_path && (getenv("DDP_HIP_FWD_LAT1") == nullptr || !p.cost_inline)) {
      if (ctx->model_h.ff) hipLaunchKernelGGL((forward_kernel_lat2<38, false, true>), dim3((unsigned)(2 * B)), dim3(128), sizeof(FwdLat2Lds<38>), ctx->stream, p);
      else hipLaunchKernelGGL((forward_kernel_lat2<38>), dim3((unsigned)(2 * B)), dim3(128), sizeof(FwdLat2Lds<38>), ctx->stream, p);
      if (!p.cost_inline) {
        hipLaunchKernelGGL((cand_cost_kernel<38>), dim3((unsigned)((B * n_alpha * d.T + 63) / 64)), dim3(64), 0, ctx->stream, p);
        hipLaunchKernelGGL(cand_sum_kernel, dim3((unsigned)((B * n_alpha + 63) / 64)), dim3(64), 0, ctx->stream, p);
      }
    } else if (lat_path) {
      const size_t lds = sizeof(double) * (size_t)(rbd::ABA_LDS_SLOTS * 38 * 8 + 8 * (76 + 76 + 38 + 38));
      hipLaunchKernelGGL((forward_kernel_lat<38>), dim3((unsigned)B), dim3(64), lds, ctx->stream, p);
    } else {
#define CALL(NJ) hipLaunchKernelGGL((forward_kernel<NJ>), dim3(grid), dim3(bs), 0, ctx->stream, p)
      DISPATCH_NJ(d.nv, CALL);
#undef CALL
    }
    prof_end(ctx, DDP_HIP_K_FWD_ROLLOUT);
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)B), dim3(256), 0, ctx->stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(state.data(), ctx->fw_state_d, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    bool searching = false;
    for (int64_t b = 0; b < B; ++b) { searching |= state[(size_t)b] == 0; floor_hit |= state[(size_t)b] == 2; }
    if (!searching) break;
  }
  HIP_TRY(hipMemcpyAsync(step_out, ctx->step_d, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
  if (dcost_out)
    HIP_TRY(hipMemcpyAsync(dcost_out, ctx->fw_dcost_acc_d, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return floor_hit ? DDP_HIP_EV_LINESEARCH_FLOOR : DDP_HIP_OK;
}
