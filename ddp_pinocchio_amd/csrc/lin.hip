// lin.hip -- linearisation of cost, dynamics and constraints along the resident trajectory.
//
// Replaces problem_t::compute_derivatives (problem.hpp:956-998; the print-only self check :999-1139
// is not reproduced).  Every (instance, t, perturbation) is an independent forward-dynamics
// evaluation, so the stencils of
//   - the first order:  forward differences, eps = sqrt(DBL_EPSILON), perturbing with integrate_x /
//     integrate_u exactly like problem.hpp:105-126 (the reference takes Pinocchio's analytic ABA
//     derivatives here, problem.hpp:463-503; the closed-form pendulum keeps its analytic partials);
//   - the second order: finite_diff_hessian_compute mode 2 (problem.hpp:152-298, eps = eps_mach^(1/4))
//     or mode 1 (problem.hpp:67-150, for models with analytic first order)
// are laid out one evaluation per lane.
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include "internal.h"
#include "lin_common.h"
#include "rbd.h"
#include "rbd_deriv.h"

namespace {



// cost derivatives, problem.hpp:958-959,982-987:  lx = 0, lxx = 0, lux = 0, lu = c u^T, luu = c I
__global__ void lin_cost_kernel(LinParams p) {
  const int64_t T = p.d.T;
  const int64_t bt = blockIdx.x;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const int n = (int)p.d.n, m = (int)p.d.m;
  const double c = p.model->c;
  const int tid = threadIdx.x;
  if (t == 0) {
    for (int i = tid; i < n; i += blockDim.x) p.lfx[(int64_t)b * n + i] = 0.0;
    for (int i = tid; i < n * n; i += blockDim.x) p.lfxx[(int64_t)b * n * n + i] = 0.0;
  }
  for (int i = tid; i < n; i += blockDim.x) p.lx[bt * n + i] = 0.0;
  for (int i = tid; i < n * n; i += blockDim.x) p.lxx[bt * n * n + i] = 0.0;
  for (int i = tid; i < m * n; i += blockDim.x) p.lux[bt * m * n + i] = 0.0;
  for (int i = tid; i < m; i += blockDim.x) p.lu[bt * m + i] = c * p.u[bt * m + i];
  for (int i = tid; i < m * m; i += blockDim.x) p.luu[bt * m * m + i] = (i % m == i / m) ? 1.0 * c : 0.0;
}

template <int NJ>
__device__ __forceinline__ void load_xu(const LinParams& p, int b, int64_t t, double* x, double* u) {
  const int nx = (int)p.d.nx, m = (int)p.d.m;
  const double* xs = p.x + ((int64_t)b * (p.d.T + 1) + t) * nx;
  const double* us = p.u + ((int64_t)b * p.d.T + t) * m;
  for (int i = 0; i < nx; ++i) x[i] = xs[i];
  for (int i = 0; i < m; ++i) u[i] = us[i];
}

// f(x, u) at the base point (+ the analytic first order of the pendulum, problem.hpp:463-503 with
// pendulum_model.hpp:116-130)
template <int NJ>
__global__ void lin_base_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  if (gid >= p.d.batch * T) return;
  const int b = (int)(gid / T);
  const int64_t t = gid % T;
  const DevModel& m = *p.model;
  const int nx = (int)p.d.nx;
  double x[2 * NJ + 1], u[NJ], f[2 * NJ + 1];
  load_xu<NJ>(p, b, t, x, u);
  rbd::eval_f<NJ>(m, x, u, f);
  for (int i = 0; i < nx; ++i) p.f_val[gid * nx + i] = f[i];
  if (!m.first_order_fd && m.kind == DDP_HIP_MODEL_PENDULUM) {
    double* fx = p.fx + gid * 4;
    double* fu = p.fu + gid * 2;
    const double aq = -9.81 / m.length * cos(x[0]);
    fx[0] = 1.0;
    fx[2] = 1.0 * m.dt;
    fx[1] = aq * m.dt;
    fx[3] = 0.0 * m.dt + 1.0;
    fu[0] = 0.0;
    fu[1] = (1.0 / m.mass) * m.dt;
  }
}

// forward-difference column j of [f_x | f_u]
template <int NJ>
__global__ void lin_first_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const int n = (int)p.d.n, mm = (int)p.d.m;
  const int W = n + mm;
  if (gid >= p.d.batch * T * W) return;
  const int j = (int)(gid % W);
  const int64_t bt = gid / W;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const DevModel& m = *p.model;
  double x[2 * NJ + 1], u[NJ], f[2 * NJ + 1];
  load_xu<NJ>(p, b, t, x, u);
  const double eps = sqrt(DBL_EPSILON);
  if (j < n) lie::perturb_x(m, x, j, eps); else u[j - n] = u[j - n] + eps;      // integrate_x / integrate_u (problem.hpp:107,117-118)
  rbd::eval_f<NJ>(m, x, u, f);
  const double* f0 = p.f_val + bt * p.d.nx;
  double* col = j < n ? p.fx + bt * n * n + (int64_t)j * n : p.fu + bt * n * mm + (int64_t)(j - n) * n;
  if (m.ff) {
    double df[2 * NJ];
    lie::difference_x(m, f0, f, df);                                             // difference_out on the group
    for (int k = 0; k < n; ++k) col[k] = df[k] / eps;
  } else {
    for (int k = 0; k < n; ++k) col[k] = (f[k] - f0[k]) / eps;
  }
}

// ---- second order, mode 2 (problem.hpp:152-298) --------------------------------------------------------
// q-dependent part of the ABA for the nv+1 configurations the mode-2 stencil visits more than once:
// cfg 0 = q, cfg 1+i = q + eps e_i (eps = eps_mach^(1/4), problem.hpp:188)
template <int NJ>
__global__ void lin_qcache_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const DevModel& m = *p.model;
  const int nv = m.nv, C = nv + 1;
  if (gid >= p.d.batch * T * C) return;
  const int cfg = (int)(gid % C);
  const int64_t bt = gid / C;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  double x[2 * NJ], u[NJ];
  load_xu<NJ>(p, b, t, x, u);
  const double eps = sqrt(sqrt(DBL_EPSILON));
  if (cfg > 0) x[cfg - 1] = x[cfg - 1] + eps;
  rbd::aba_qpart<NJ>(m, x, p.qcache + (bt * C + cfg) * (int64_t)nv * rbd::QC_STRIDE);
}

// (q, v)-dependent part for the 2 nv + 1 (q, v) pairs shared by several stencil points:
// vcfg 0 = (q, v), 1+i = (q, v + eps e_i), nv+1+i = (q + eps e_i, v)
template <int NJ>
__global__ void lin_vcache_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const DevModel& m = *p.model;
  const int nv = m.nv, C = 2 * nv + 1;
  if (gid >= p.d.batch * T * C) return;
  const int vcfg = (int)(gid % C);
  const int64_t bt = gid / C;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  double x[2 * NJ], u[NJ];
  load_xu<NJ>(p, b, t, x, u);
  const double eps = sqrt(sqrt(DBL_EPSILON));
  if (vcfg >= 1 && vcfg <= nv) x[nv + vcfg - 1] = x[nv + vcfg - 1] + eps;
  const int cfg = vcfg > nv ? vcfg - nv : 0;
  rbd::aba_vpart_cached<NJ>(m, p.qcache + (bt * (nv + 1) + cfg) * (int64_t)nv * rbd::QC_STRIDE, x + nv,
                            p.vcache + (bt * C + vcfg) * (int64_t)nv * rbd::VC_STRIDE);
}

// diagonal entries, :192-222
template <int NJ>
__global__ void lin_diag_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const int n = (int)p.d.n, mm = (int)p.d.m;
  const int W = n + mm;
  if (gid >= p.d.batch * T * W) return;
  const int i = (int)(gid % W);
  const int64_t bt = gid / W;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const DevModel& m = *p.model;
  double x[2 * NJ + 1], u[NJ], f1[2 * NJ + 1];
  load_xu<NJ>(p, b, t, x, u);
  const double eps = sqrt(sqrt(DBL_EPSILON));
  const double eps2 = eps * eps;
  const bool at_x = i < n;
  const int idx = at_x ? i : i - n;
  if (at_x) lie::perturb_x(m, x, idx, eps); else u[idx] = u[idx] + eps;
  if (p.qcache) {
    const int nv = m.nv, cfg = (at_x && idx < nv) ? 1 + idx : 0;
    const double* qc = p.qcache + (bt * (nv + 1) + cfg) * (int64_t)nv * rbd::QC_STRIDE;
    if (!at_x) rbd::eval_f_ucached<NJ>(m, qc, p.vcache + (bt * (2 * nv + 1)) * (int64_t)nv * rbd::VC_STRIDE, x, u, f1);
    else rbd::eval_f_cached<NJ>(m, qc, x, u, f1);
  } else {
    rbd::eval_f<NJ>(m, x, u, f1);
  }
  const double* f0 = p.f_val + bt * p.d.nx;
  const double* fcol = at_x ? p.fx + bt * n * n + (int64_t)idx * n : p.fu + bt * n * mm + (int64_t)idx * n;
  double* tensor = at_x ? p.fxx + bt * n * n * n : p.fuu + bt * n * mm * mm;
  const int L = at_x ? n : mm;
  if (m.ff) {                       // difference_out on the group (problem.hpp:206), then the vector-space expression
    double dfv[2 * NJ];
    lie::difference_x(m, f0, f1, dfv);
    for (int k = 0; k < n; ++k) f1[k] = dfv[k];
  }
  for (int k = 0; k < n; ++k) {
    double df = m.ff ? f1[k] : f1[k] - f0[k];      // difference_out
    df -= eps * fcol[k];
    df *= 2;
    tensor[k + (int64_t)idx * n + (int64_t)idx * n * L] = df / eps2;
  }
}

// off-diagonal entries, :226-296: one lane per unordered pair i < j of the n+m directions.
// PAIRS = 0: every pair, full ABA (no caches).  PAIRS = 1: two q directions (both perturb the configuration: full
// ABA).  PAIRS = 2: (q or v, v): the q-dependent part comes from the q-cache.  PAIRS = 3: (q, v or u; u): only tau
// differs from a cached (q, v) pair, the evaluation is the force / acceleration passes alone.
template <int NJ, int PAIRS>
__global__ __launch_bounds__(LBS, (PAIRS == 1 || PAIRS == 0 ? 4 : 5)) void lin_offdiag_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const int n = (int)p.d.n, mm = (int)p.d.m;
  const int W = n + mm;
  const DevModel& m = *p.model;
  const int nv = m.nv;
  const int64_t TRI = (int64_t)nv * (nv - 1) / 2;
  const int64_t P = PAIRS == 0 ? (int64_t)W * (W - 1) / 2
                  : PAIRS == 1 ? TRI
                  : PAIRS == 2 ? (int64_t)nv * nv + TRI
                               : 2 * (int64_t)nv * nv + TRI;
  const bool valid = gid < p.d.batch * T * P;
  const int64_t pid = valid ? gid % P : 0;
  const int64_t bt = valid ? gid / P : 0;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  auto tri = [](int64_t q, int Wd, int& ii, int& jj) {      // q -> (ii, jj), ii < jj < Wd, enumerated row by row
    int a = (int)floor(((2.0 * Wd - 1.0) - sqrt((2.0 * Wd - 1.0) * (2.0 * Wd - 1.0) - 8.0 * (double)q)) * 0.5);
    if (a < 0) a = 0;
    while ((int64_t)a * (2 * Wd - a - 1) / 2 > q) --a;
    while ((int64_t)(a + 1) * (2 * Wd - a - 2) / 2 <= q) ++a;
    ii = a;
    jj = (int)(q - (int64_t)a * (2 * Wd - a - 1) / 2) + a + 1;
  };
  int i, j;
  if (PAIRS == 0) tri(pid, W, i, j);
  else if (PAIRS == 1) tri(pid, nv, i, j);
  else if (PAIRS == 2) {
    if (pid < (int64_t)nv * nv) { i = (int)(pid / nv); j = nv + (int)(pid % nv); }                 // (q_i, v_j)
    else { tri(pid - (int64_t)nv * nv, nv, i, j); i += nv; j += nv; }                              // (v_i, v_j)
  } else {
    if (pid < 2 * (int64_t)nv * nv) { i = (int)(pid / nv); j = 2 * nv + (int)(pid % nv); }         // (q_i or v_i, u_j)
    else { tri(pid - 2 * (int64_t)nv * nv, nv, i, j); i += 2 * nv; j += 2 * nv; }                  // (u_i, u_j)
  }

  const double eps = sqrt(sqrt(DBL_EPSILON));
  const double eps2 = eps * eps;
  // PAIRS == 3 keeps no private copy of x, u or f(x+dx): the control is read through a functor and the output
  // rows are formed on the fly from the accelerations
  double f1[PAIRS == 3 ? 1 : 2 * NJ + 1], qdd[PAIRS == 3 ? NJ : 1];
  const double* xg = p.x + ((int64_t)b * (T + 1) + t) * p.d.nx;
  const double* ug = p.u + ((int64_t)b * T + t) * mm;
  if (valid) {
    if (PAIRS == 3) {
      const int cfg = i < nv ? 1 + i : 0;
      const int vcfg = i < nv ? nv + 1 + i : (i < 2 * nv ? 1 + (i - nv) : 0);
      const int iu = i - n, ju = j - n;
      rbd::aba_u_cached<NJ>(m, p.qcache + (bt * (nv + 1) + cfg) * (int64_t)nv * rbd::QC_STRIDE,
                            p.vcache + (bt * (2 * nv + 1) + vcfg) * (int64_t)nv * rbd::VC_STRIDE,
                            [&](int k) { double v = ug[k]; if (k == iu) v = v + eps; if (k == ju) v = v + eps; return v; }, qdd);
    } else {
      double x[2 * NJ + 1], u[NJ];
      load_xu<NJ>(p, b, t, x, u);
      // both directions in ONE step of the group (integrate_x of dx = eps e_i + eps e_j, problem.hpp:262-263): on a vector
      // space the two additions commute; on SE(3) the base twist eps (e_i + e_j) is integrated once
      if (PAIRS == 0 && m.ff && i < 6 && j < 6) {
        double nu[6] = {0, 0, 0, 0, 0, 0}, q7[7];
        nu[i] = eps; nu[j] = eps;
        lie::se3_integrate(x, nu, q7);
        for (int k = 0; k < 7; ++k) x[k] = q7[k];
      } else {
        if (i < n) lie::perturb_x(m, x, i, eps); else u[i - n] = u[i - n] + eps;
        if (j < n) lie::perturb_x(m, x, j, eps); else u[j - n] = u[j - n] + eps;
      }
      if (PAIRS == 2) {
        const int cfg = i < nv ? 1 + i : 0;
        rbd::eval_f_cached<NJ>(m, p.qcache + (bt * (nv + 1) + cfg) * (int64_t)nv * rbd::QC_STRIDE, x, u, f1);
      } else {
        rbd::eval_f<NJ>(m, x, u, f1);
        if (PAIRS == 0 && m.ff) {     // difference_out on the group (problem.hpp:268); the output stage then subtracts nothing
          double dfv[2 * NJ];
          lie::difference_x(m, p.f_val + bt * p.d.nx, f1, dfv);
          for (int k = 0; k < n; ++k) f1[k] = dfv[k];
        }
      }
    }
  }
  // row k of f(x + dx, u + du) (dynamics_t::eval_to, problem.hpp:441-461)
  auto f1_at = [&](int k) -> double {
    if (PAIRS != 3) return f1[k];
    if (k < nv) {
      const double xq = k == i ? xg[k] + eps : xg[k];
      const double xv = (nv + k) == i ? xg[nv + k] + eps : xg[nv + k];
      const double vo = m.dt * xv;
      return xq + vo;
    }
    const double xv = k == i ? xg[k] + eps : xg[k];
    return xv + qdd[k - nv] * m.dt;
  };

  // Output stage.  Each stencil point owns one 76-double column of a tensor (and reads five more columns);
  // done lane-per-point that is 64 scattered 8-byte accesses per instruction.  Instead the wave transposes the
  // f(x+dx) values through LDS, 16 rows at a time, and walks the points four at a time with 16 lanes on each
  // column: every access is four 128-byte runs.
  constexpr int CH = 16, EPI = LBS / CH;               // rows per chunk, points per instruction
  __shared__ double s_f[CH][LBS];
  __shared__ int s_i[LBS], s_j[LBS];
  __shared__ int64_t s_bt[LBS];
  const int lane = threadIdx.x;
  s_i[lane] = valid ? i : -1;
  s_j[lane] = j;
  s_bt[lane] = bt;
  const int kk = lane % CH, esub = lane / CH;
  for (int c0 = 0; c0 < n; c0 += CH) {
#pragma unroll
    for (int q = 0; q < CH; ++q)
      if (c0 + q < n && valid) s_f[q][lane] = f1_at(c0 + q);
    __syncthreads();
    const int k = c0 + kk;
    for (int r = 0; r < LBS / EPI; ++r) {
      const int e = r * EPI + esub;
      const int ie = s_i[e];
      if (ie < 0 || k >= n) continue;
      const int je = s_j[e];
      const int64_t bte = s_bt[e];
      const bool at_x_1 = ie < n, at_x_2 = je < n;
      const int idx_1 = at_x_1 ? ie : ie - n, idx_2 = at_x_2 ? je : je - n;
      const double* f0 = p.f_val + bte * p.d.nx;
      double* fxx = p.fxx + bte * n * n * n;
      double* fux = p.fux + bte * n * mm * n;
      double* fuu = p.fuu + bte * n * mm * mm;
      const double* fcol_1 = at_x_1 ? p.fx + bte * n * n + (int64_t)idx_1 * n : p.fu + bte * n * mm + (int64_t)idx_1 * n;
      const double* fcol_2 = at_x_2 ? p.fx + bte * n * n + (int64_t)idx_2 * n : p.fu + bte * n * mm + (int64_t)idx_2 * n;
      const double* tensor_1 = at_x_1 ? fxx : fuu;
      const double* tensor_2 = at_x_2 ? fxx : fuu;
      const int L1 = at_x_1 ? n : mm, L2 = at_x_2 ? n : mm;
      double* tensor;
      int L;
      if (at_x_1) { if (at_x_2) { tensor = fxx; L = n; } else { tensor = fux; L = mm; } }
      else { tensor = fuu; L = mm; }
      double df = (PAIRS == 0 && m.ff) ? s_f[kk][e] : s_f[kk][e] - f0[k];   // difference_out
      df -= eps * fcol_1[k];
      df -= eps * fcol_2[k];
      df *= 2;
      const double val = 0.5 * (df / eps2 - tensor_1[k + (int64_t)idx_1 * n + (int64_t)idx_1 * n * L1] -
                                tensor_2[k + (int64_t)idx_2 * n + (int64_t)idx_2 * n * L2]);
      tensor[k + (int64_t)idx_2 * n + (int64_t)idx_1 * n * L] = val;
      if (at_x_1 == at_x_2) tensor[k + (int64_t)idx_1 * n + (int64_t)idx_2 * n * L] = val;
    }
    __syncthreads();
  }
}

// ---- small-model paths (nv <= 6): first order as a device function, used by the constraint chain and mode 1
template <int NJ>
__device__ void first_order_f(const DevModel& m, const double* x, const double* u, double* fx, double* fu, double* f) {
  const int nv = m.nv, n = 2 * nv, mm = nv;
  if (!m.first_order_fd && m.kind == DDP_HIP_MODEL_TREE) {
    // analytic, as the reference: d_dynamics_aba (problem.hpp:495)
    if constexpr (NJ <= 6) rbdd::first_order_analytic_lane<NJ>(m, x, u, fx, fu, f);
    return;
  }
  rbd::eval_f<NJ>(m, x, u, f);
  if (!m.first_order_fd) {
    const double aq = -9.81 / m.length * cos(x[0]);
    fx[0] = 1.0; fx[2] = 1.0 * m.dt; fx[1] = aq * m.dt; fx[3] = 0.0 * m.dt + 1.0;
    fu[0] = 0.0; fu[1] = (1.0 / m.mass) * m.dt;
    return;
  }
  const double eps = sqrt(DBL_EPSILON);
  double xp[2 * NJ], up[NJ], fp[2 * NJ];
  for (int j = 0; j < n + mm; ++j) {
    for (int k = 0; k < n; ++k) xp[k] = x[k];
    for (int k = 0; k < mm; ++k) up[k] = u[k];
    if (j < n) xp[j] = x[j] + eps; else up[j - n] = u[j - n] + eps;
    rbd::eval_f<NJ>(m, xp, up, fp);
    double* col = j < n ? fx + j * n : fu + (j - n) * n;
    for (int k = 0; k < n; ++k) col[k] = (fp[k] - f[k]) / eps;
  }
}

// analytic first order of a small tree model (nv <= 6), one lane per (instance, t)
template <int NJ>
__global__ void lin_first_analytic_small_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  if (gid >= p.d.batch * T) return;
  const int b = (int)(gid / T);
  const int64_t t = gid % T;
  const DevModel& m = *p.model;
  const int n = 2 * m.nv, mm = m.nv;
  double x[2 * NJ], u[NJ], f[2 * NJ];
  load_xu<NJ>(p, b, t, x, u);
  first_order_f<NJ>(m, x, u, p.fx + gid * n * n, p.fu + gid * n * mm, f);     // f_val itself is lin_base_kernel's (same arithmetic)
}

// constraint value through the advance chain (problem.hpp:563-567)
template <int NJ>
__device__ void eq_eval(const DevModel& m, const double* target, int e, const double* x, const double* u, double* out) {
  const int nx = m.nq + m.nv;
  double xa[2 * NJ + 1], xb[2 * NJ + 1];
  for (int i = 0; i < nx; ++i) xa[i] = x[i];
  for (int k = 0; k < m.eq_advance; ++k) {
    if (k + 1 < m.eq_advance) rbd::eval_f<NJ>(m, xa, u, xb);
    else { for (int i = m.nq; i < nx; ++i) xb[i] = xa[i]; rbd::eval_f_q<NJ>(m, xa, xb); }   // the constraint reads q only: no dynamics in the last step
    for (int i = 0; i < nx; ++i) xa[i] = xb[i];
  }
  if (m.eq_kind == DDP_HIP_EQ_CONFIG) {
    for (int i = 0; i < e; ++i) out[i] = xa[i] - target[i];
  } else {
    double pos[3];
    rbd::frame_position<NJ>(m, xa, pos, nullptr);
    for (int i = 0; i < e; ++i) out[i] = pos[i] - target[i];
  }
}

// constraint first order through the advance chain: constraint_advance_time_t::first_order_deriv,
// problem.hpp:569-605 (out_x = eq_n_x * fx_n, out_u = eq_n_x * fu_n; the inner eq_n_u is dropped as the
// reference asserts it to be zero), around config_constraint_t :808-845 / spatial_constraint_t :691-722
template <int NJ, int ADV>
__device__ void eq_first_order(const DevModel& m, const double* target, int e, const double* x, const double* u,
                               double* out_x, double* out_u, double* out) {
  constexpr int N = 2 * NJ, M = NJ, EM = NJ > 3 ? NJ : 3;
  const int nv = m.nv, n = 2 * nv, mm = nv;
  double Fx[ADV > 0 ? ADV : 1][N * N], Fu[ADV > 0 ? ADV : 1][N * M];
  double xa[N], xb[N];
  for (int i = 0; i < n; ++i) xa[i] = x[i];
  const int adv = m.eq_advance;
  for (int k = 0; k < adv; ++k) {
    first_order_f<NJ>(m, xa, u, Fx[k], Fu[k], xb);
    for (int i = 0; i < n; ++i) xa[i] = xb[i];
  }
  double ex[EM * N], tmp[EM * N];
  for (int i = 0; i < e * n; ++i) ex[i] = 0.0;
  if (m.eq_kind == DDP_HIP_EQ_CONFIG) {
    for (int i = 0; i < e; ++i) { out[i] = xa[i] - target[i]; ex[i + i * e] = 1.0; }   // d_difference_dq_finish = I
  } else {
    double pos[3], J[3 * NJ];
    rbd::frame_position<NJ>(m, xa, pos, J);
    for (int i = 0; i < e; ++i) out[i] = pos[i] - target[i];
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < e; ++i) ex[i + j * e] = J[i + 3 * j];
  }
  for (int i = 0; i < e * mm; ++i) out_u[i] = 0.0;
  for (int k = adv - 1; k >= 0; --k) {
    for (int j = 0; j < mm; ++j)
      for (int i = 0; i < e; ++i) {
        double s = 0;
        for (int l = 0; l < n; ++l) s += ex[i + l * e] * Fu[k][l + j * n];
        out_u[i + j * e] = s;
      }
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < e; ++i) {
        double s = 0;
        for (int l = 0; l < n; ++l) s += ex[i + l * e] * Fx[k][l + j * n];
        tmp[i + j * e] = s;
      }
    for (int i = 0; i < e * n; ++i) ex[i] = tmp[i];
  }
  for (int i = 0; i < e * n; ++i) out_x[i] = ex[i];
}

// ---- constraint chain, large models: the same chain rule as eq_first_order, as three kernels ---------------------
// (the per-lane variant above would need n x n private matrices per lane)
template <int NJ>
__global__ void eq_chain_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  if (gid >= p.d.batch * T) return;
  const int b = (int)(gid / T);
  const int64_t t = gid % T;
  const int e = (int)p.ne[t];
  if (e == 0) return;
  const DevModel& m = *p.model;
  const int nv = m.nv, n = 2 * nv, K = m.eq_advance, nx = (int)p.d.nx;
  const int64_t Eo = p.Epre[t], Eb = (int64_t)b * p.d.Etot + Eo;
  double xa[2 * NJ + 1], xb[2 * NJ + 1], u[NJ];
  load_xu<NJ>(p, b, t, xa, u);
  for (int k = 0; k < K; ++k) {                          // x_{k+1} = f(x_k, u): the SAME u at every look-ahead step
    if (k + 1 < K || m.ff) rbd::eval_f<NJ>(m, xa, u, xb);   // (free flyer: eq_fdjac differences whole states on the group)
    else { for (int i = m.nq; i < nx; ++i) xb[i] = xa[i]; rbd::eval_f_q<NJ>(m, xa, xb); }   // x_K: only its configuration is read (the constraint, and eq_fdjac's
    for (int i = 0; i < nx; ++i) { xa[i] = xb[i]; p.eq_xk[(gid * K + k) * nx + i] = xb[i]; }  // q rows below) -- its velocity half is a placeholder
  }
  double* C = p.eq_c + gid * (int64_t)p.d.emax * n;
  for (int i = 0; i < e * n; ++i) C[i] = 0.0;
  const double* target = p.target + Eo;
  if (m.eq_kind == DDP_HIP_EQ_CONFIG) {
    for (int i = 0; i < e; ++i) { p.eq_val[Eb + i] = xa[i] - target[i]; C[i + i * e] = 1.0; }   // d_difference_dq_finish = I
  } else {
    double pos[3], J[3 * NJ];
    rbd::frame_position<NJ>(m, xa, pos, J);
    for (int i = 0; i < e; ++i) p.eq_val[Eb + i] = pos[i] - target[i];
    for (int j = 0; j < nv; ++j)
      for (int i = 0; i < e; ++i) C[i + j * e] = J[i + 3 * j];
  }
}

// forward-difference f_x at the look-ahead states x_1 .. x_{K-1}: one lane per (b, t, k, column)
template <int NJ>
__global__ void eq_fdjac_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const DevModel& m = *p.model;
  const int n = 2 * m.nv, K = m.eq_advance;
  if (K < 2 || gid >= p.d.batch * T * (K - 1) * n) return;
  const int j = (int)(gid % n);
  const int k = (int)((gid / n) % (K - 1));               // Jacobian at x_{k+1}
  const int64_t bt = gid / ((int64_t)n * (K - 1));
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  if (p.ne[t] == 0) return;
  double x[2 * NJ + 1], u[NJ], f[2 * NJ + 1];
  const int nx = (int)p.d.nx;
  const double* xk = p.eq_xk + (bt * K + k) * nx;          // x_{k+1}
  const double* xk1 = p.eq_xk + (bt * K + k + 1) * nx;     // x_{k+2} = f(x_{k+1}, u)
  const double* us = p.u + ((int64_t)b * T + t) * m.nv;
  for (int i = 0; i < nx; ++i) x[i] = xk[i];
  for (int i = 0; i < m.nv; ++i) u[i] = us[i];
  const double eps = sqrt(DBL_EPSILON);
  lie::perturb_x(m, x, j, eps);
  double* col = p.eq_fxk + (bt * (K - 1) + k) * (int64_t)n * n + (int64_t)j * n;
  if (k == K - 2 && !m.ff) {
    // f_x(x_{K-1}) is multiplied from the left by the base jacobian C = [C_q | 0] (both constraint kinds read q only) and by
    // nothing else: only its q rows matter, and those difference q+ = q + dt v -- no dynamics.  The very same values as the
    // full column's q rows (the v rows, multiplied by exact zeros in eq_combine, are written as zeros): 76 forward-dynamics
    // evaluations per constrained (instance, t) less.
    const int nv = m.nv;
    for (int i = 0; i < nv; ++i) { const double vo = m.dt * x[nv + i]; const double fq = x[i] + vo; col[i] = (fq - xk1[i]) / eps; }
    for (int i = nv; i < n; ++i) col[i] = 0.0;
    return;
  }
  rbd::eval_f<NJ>(m, x, u, f);
  if (m.ff) {
    double df[2 * NJ];
    lie::difference_x(m, xk1, f, df);
    for (int i = 0; i < n; ++i) col[i] = df[i] / eps;
  } else {
    for (int i = 0; i < n; ++i) col[i] = (f[i] - xk1[i]) / eps;
  }
}

// eq_x = C f_x(x_{K-1}) ... f_x(x_1) f_x(x_0),  eq_u = C f_x(x_{K-1}) ... f_x(x_1) f_u(x_0)   (problem.hpp:603-604)
__global__ void eq_combine_kernel(LinParams p) {
  const int64_t bt = blockIdx.x;
  const int64_t T = p.d.T;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const int e = (int)p.ne[t];
  if (e == 0) return;
  const int n = (int)p.d.n, mm = (int)p.d.m, K = p.model->eq_advance;
  const int64_t Eb = (int64_t)b * p.d.Etot + p.Epre[t];
  extern __shared__ double sm[];
  double* ex = sm;
  double* tmp = sm + (int64_t)p.d.emax * n;
  const double* C = p.eq_c + bt * (int64_t)p.d.emax * n;
  for (int i = threadIdx.x; i < e * n; i += blockDim.x) ex[i] = C[i];
  __syncthreads();
  for (int k = K - 2; k >= 0; --k) {
    const double* F = p.eq_fxk + (bt * (K - 1) + k) * (int64_t)n * n;
    for (int idx = threadIdx.x; idx < e * n; idx += blockDim.x) {
      const int i = idx % e, j = idx / e;
      double s = 0;
      for (int l = 0; l < n; ++l) s += ex[i + l * e] * F[l + (int64_t)j * n];
      tmp[idx] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < e * n; i += blockDim.x) ex[i] = tmp[i];
    __syncthreads();
  }
  const double* fx = p.fx + bt * n * n;
  const double* fu = p.fu + bt * n * mm;
  for (int idx = threadIdx.x; idx < e * n; idx += blockDim.x) {
    const int i = idx % e, j = idx / e;
    double s = 0;
    for (int l = 0; l < n; ++l) s += ex[i + l * e] * fx[l + j * n];
    p.eq_x[Eb * n + idx] = s;
  }
  for (int idx = threadIdx.x; idx < e * mm; idx += blockDim.x) {
    const int i = idx % e, j = idx / e;
    double s = 0;
    for (int l = 0; l < n; ++l) s += ex[i + l * e] * fu[l + j * n];
    p.eq_u[Eb * mm + idx] = s;
  }
}

#define MAXADV 4

template <int NJ>
__global__ void eq_first_kernel(LinParams p) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  if (gid >= p.d.batch * T) return;
  const int b = (int)(gid / T);
  const int64_t t = gid % T;
  const int e = (int)p.ne[t];
  if (e == 0) return;
  const DevModel& m = *p.model;
  const int n = (int)p.d.n, mm = (int)p.d.m;
  const int64_t Eo = p.Epre[t], Eb = (int64_t)b * p.d.Etot + Eo;
  double x[2 * NJ], u[NJ];
  load_xu<NJ>(p, b, t, x, u);
  eq_first_order<NJ, MAXADV>(m, p.target + Eo, e, x, u, p.eq_x + Eb * n, p.eq_u + Eb * mm, p.eq_val + Eb);
}

// mode 1 (problem.hpp:67-150) for the dynamics (analytic first order only) and for the constraint chain
template <int NJ>
__global__ void second_m1_kernel(LinParams p, int is_eq) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const int n = (int)p.d.n, mm = (int)p.d.m;
  const int W = n + mm;
  if (gid >= p.d.batch * T * W) return;
  const int i = (int)(gid % W);
  const int64_t bt = gid / W;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const DevModel& m = *p.model;
  constexpr int N = 2 * NJ, M = NJ, EM = NJ > 3 ? NJ : 3;
  double x[N], u[M];
  load_xu<NJ>(p, b, t, x, u);
  const double eps = sqrt(DBL_EPSILON);
  const bool at_x = i < n;
  const int idx = at_x ? i : i - n;
  if (at_x) x[idx] = x[idx] + eps; else u[idx] = u[idx] + eps;
  if (!is_eq) {
    double fx_[N * N], fu_[N * M], out_[N];
    first_order_f<NJ>(m, x, u, fx_, fu_, out_);
    const double* ox = p.fx + bt * n * n;
    const double* ou = p.fu + bt * n * mm;
    const int o = n;
    if (at_x) {
      for (int k = 0; k < o; ++k) {
        for (int j = 0; j < n; ++j) p.fxx[bt * n * n * n + k + (int64_t)j * o + (int64_t)idx * o * n] = (fx_[k + j * o] - ox[k + j * o]) / eps;
        for (int j = 0; j < mm; ++j) p.fux[bt * n * mm * n + k + (int64_t)j * o + (int64_t)idx * o * mm] = (fu_[k + j * o] - ou[k + j * o]) / eps;
      }
    } else {
      for (int k = 0; k < o; ++k)
        for (int j = 0; j < mm; ++j) p.fuu[bt * n * mm * mm + k + (int64_t)j * o + (int64_t)idx * o * mm] = (fu_[k + j * o] - ou[k + j * o]) / eps;
    }
  } else {
    const int e = (int)p.ne[t];
    if (e == 0) return;
    const int64_t Eo = p.Epre[t], Eb = (int64_t)b * p.d.Etot + Eo;
    double ex_[EM * N], eu_[EM * M], out_[EM];
    eq_first_order<NJ, MAXADV>(m, p.target + Eo, e, x, u, ex_, eu_, out_);
    const double* ox = p.eq_x + Eb * n;
    const double* ou = p.eq_u + Eb * mm;
    const int o = e;
    if (at_x) {
      for (int k = 0; k < o; ++k) {
        for (int j = 0; j < n; ++j) p.eq_xx[Eb * n * n + k + (int64_t)j * o + (int64_t)idx * o * n] = (ex_[k + j * o] - ox[k + j * o]) / eps;
        for (int j = 0; j < mm; ++j) p.eq_ux[Eb * mm * n + k + (int64_t)j * o + (int64_t)idx * o * mm] = (eu_[k + j * o] - ou[k + j * o]) / eps;
      }
    } else {
      for (int k = 0; k < o; ++k)
        for (int j = 0; j < mm; ++j) p.eq_uu[Eb * mm * mm + k + (int64_t)j * o + (int64_t)idx * o * mm] = (eu_[k + j * o] - ou[k + j * o]) / eps;
    }
  }
}

// mode 2 for the constraint chain: stage 0 = diagonal, stage 1 = off-diagonal
template <int NJ>
__global__ void eq_second_m2_kernel(LinParams p, int stage) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t T = p.d.T;
  const int n = (int)p.d.n, mm = (int)p.d.m;
  const int W = n + mm;
  const int64_t P = stage == 0 ? W : (int64_t)W * (W - 1) / 2;
  if (gid >= p.d.batch * T * P) return;
  const int64_t pid = gid % P;
  const int64_t bt = gid / P;
  const int b = (int)(bt / T);
  const int64_t t = bt % T;
  const int e = (int)p.ne[t];
  if (e == 0) return;
  int i, j;
  if (stage == 0) { i = (int)pid; j = -1; }
  else {
    i = (int)floor(((2.0 * W - 1.0) - sqrt((2.0 * W - 1.0) * (2.0 * W - 1.0) - 8.0 * (double)pid)) * 0.5);
    if (i < 0) i = 0;
    while ((int64_t)i * (2 * W - i - 1) / 2 > pid) --i;
    while ((int64_t)(i + 1) * (2 * W - i - 2) / 2 <= pid) ++i;
    j = (int)(pid - (int64_t)i * (2 * W - i - 1) / 2) + i + 1;
  }
  const DevModel& m = *p.model;
  constexpr int EM = NJ > 3 ? NJ : 3;
  double x[2 * NJ + 1], u[NJ], f1[EM];
  load_xu<NJ>(p, b, t, x, u);
  const double eps = sqrt(sqrt(DBL_EPSILON));
  const double eps2 = eps * eps;
  const int64_t Eo = p.Epre[t], Eb = (int64_t)b * p.d.Etot + Eo;
  const double* f0 = p.eq_val + Eb;
  double* exx = p.eq_xx + Eb * n * n;
  double* eux = p.eq_ux + Eb * mm * n;
  double* euu = p.eq_uu + Eb * mm * mm;
  const bool at_x_1 = i < n;
  const int idx_1 = at_x_1 ? i : i - n;
  const bool both_base = stage != 0 && m.ff && i < 6 && j < 6;     // one step of the group for a pair of base directions
  if (both_base) {
    double nu[6] = {0, 0, 0, 0, 0, 0}, q7[7];
    nu[i] = eps; nu[j] = eps;
    lie::se3_integrate(x, nu, q7);
    for (int k = 0; k < 7; ++k) x[k] = q7[k];
  } else if (at_x_1) lie::perturb_x(m, x, idx_1, eps); else u[idx_1] = u[idx_1] + eps;
  const double* fcol_1 = at_x_1 ? p.eq_x + Eb * n + (int64_t)idx_1 * e : p.eq_u + Eb * mm + (int64_t)idx_1 * e;
  const int L1 = at_x_1 ? n : mm;
  double* tensor_1 = at_x_1 ? exx : euu;
  if (stage == 0) {
    eq_eval<NJ>(m, p.target + Eo, e, x, u, f1);
    for (int k = 0; k < e; ++k) {
      double df = f1[k] - f0[k];
      df -= eps * fcol_1[k];
      df *= 2;
      tensor_1[k + (int64_t)idx_1 * e + (int64_t)idx_1 * e * L1] = df / eps2;
    }
    return;
  }
  const bool at_x_2 = j < n;
  const int idx_2 = at_x_2 ? j : j - n;
  if (both_base) {} else if (at_x_2) lie::perturb_x(m, x, idx_2, eps); else u[idx_2] = u[idx_2] + eps;
  const double* fcol_2 = at_x_2 ? p.eq_x + Eb * n + (int64_t)idx_2 * e : p.eq_u + Eb * mm + (int64_t)idx_2 * e;
  const int L2 = at_x_2 ? n : mm;
  const double* tensor_2 = at_x_2 ? exx : euu;
  double* tensor;
  int L;
  if (at_x_1) { if (at_x_2) { tensor = exx; L = n; } else { tensor = eux; L = mm; } }
  else { tensor = euu; L = mm; }
  eq_eval<NJ>(m, p.target + Eo, e, x, u, f1);
  for (int k = 0; k < e; ++k) {
    double df = f1[k] - f0[k];
    df -= eps * fcol_1[k];
    df -= eps * fcol_2[k];
    df *= 2;
    const double val = 0.5 * (df / eps2 - tensor_1[k + (int64_t)idx_1 * e + (int64_t)idx_1 * e * L1] -
                              tensor_2[k + (int64_t)idx_2 * e + (int64_t)idx_2 * e * L2]);
    tensor[k + (int64_t)idx_2 * e + (int64_t)idx_1 * e * L] = val;
    if (at_x_1 == at_x_2) tensor[k + (int64_t)idx_1 * e + (int64_t)idx_2 * e * L] = val;
  }
}

// T(:, i, j) = T(:, j, i) for i < j: the mirror images of a symmetric tensor's entries (f_xx: L = n; f_uu: L = m), which the
// static stencil leaves out while the backward sweep is known not to read them (LinParams::skip_qv_mirror); formed when
// somebody else asks for FXX / FUU
__global__ void tensor_mirror_kernel(double* Tn, int64_t BT, int n, int L) {
  const int64_t per = (int64_t)n * L * L, total = BT * per;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bt = g / per, r = g % per;
    const int k = (int)(r % n), j = (int)((r / n) % L), c = (int)(r / ((int64_t)n * L));   // entry (k, j, c): column j of slab c
    if (j >= c) continue;
    double* T = Tn + bt * per;
    T[k + (int64_t)j * n + (int64_t)c * n * L] = T[k + (int64_t)c * n + (int64_t)j * n * L];   // its direct twin: column c of slab j
  }
}

// rows k < nv of every column of a tensor (O = n rows, `cols` columns per (instance, t)) to zero: what the static stencil
// relies on when it skips them (LinParams::skip_top)
__global__ void tensor_zero_top_kernel(double* Tn, int64_t total_cols, int n, int nv) {
  const int64_t total = total_cols * nv;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = g / nv;
    Tn[col * n + (g % nv)] = 0.0;
  }
}

LinParams make_params(ddp_hip_ctx* ctx) {
  LinParams p{};
  p.d = ctx->d;
  p.model = ctx->model_d;
  p.ne = ctx->ne_d;
  p.Epre = ctx->Epre_d;
  p.target = ctx->target_d;
  auto S = [&](int s) { return ctx->seq[s].ptr; };
  p.x = S(DDP_HIP_SEQ_X); p.u = S(DDP_HIP_SEQ_U);
  p.lfx = S(DDP_HIP_SEQ_LFX); p.lfxx = S(DDP_HIP_SEQ_LFXX);
  p.lx = S(DDP_HIP_SEQ_LX); p.lu = S(DDP_HIP_SEQ_LU); p.lxx = S(DDP_HIP_SEQ_LXX); p.lux = S(DDP_HIP_SEQ_LUX); p.luu = S(DDP_HIP_SEQ_LUU);
  p.f_val = S(DDP_HIP_SEQ_F_VAL); p.fx = S(DDP_HIP_SEQ_FX); p.fu = S(DDP_HIP_SEQ_FU);
  p.fxx = S(DDP_HIP_SEQ_FXX); p.fux = S(DDP_HIP_SEQ_FUX); p.fuu = S(DDP_HIP_SEQ_FUU);
  p.eq_val = S(DDP_HIP_SEQ_EQ_VAL); p.eq_x = S(DDP_HIP_SEQ_EQ_X); p.eq_u = S(DDP_HIP_SEQ_EQ_U);
  p.eq_xx = S(DDP_HIP_SEQ_EQ_XX); p.eq_ux = S(DDP_HIP_SEQ_EQ_UX); p.eq_uu = S(DDP_HIP_SEQ_EQ_UU);
  p.has_tensors = (ctx->flags & DDP_HIP_FLAG_NO_TENSORS) ? 0 : 1;
  p.skip_top = (ctx->lin_static && ctx->model_h.fd_mode == 2 && !ctx->model_h.ff && getenv("DDP_HIP_FXX_FULL") == nullptr) ? 1 : 0;
  p.skip_qv_mirror = (ctx->jobs_sym_d && ctx->lin_static && ctx->model_h.fd_mode == 2 && getenv("DDP_HIP_K3_NO_SYM") == nullptr &&
                      getenv("DDP_HIP_FXX_FULL") == nullptr) ? 1 : 0;
  p.eq_xk = ctx->eq_ws;
  if (p.eq_xk) {
    const Dims& dd = ctx->d;
    const int64_t K = ctx->model_h.eq_advance;
    p.eq_fxk = p.eq_xk + dd.batch * dd.T * K * dd.nx;
    p.eq_c = p.eq_fxk + dd.batch * dd.T * (K > 1 ? K - 1 : 0) * dd.n * dd.n;
  }
  p.qcache = reinterpret_cast<double*>(ctx->lin_ws);
  p.ncfg = ctx->lin_ncfg; p.nvcfg = ctx->lin_nvcfg;
  p.vcache = p.qcache ? p.qcache + ctx->d.batch * ctx->d.T * (int64_t)ctx->lin_ncfg * ctx->d.nv * rbd::QC_STRIDE : nullptr;
  return p;
}

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + LBS - 1) / LBS); }

template <int NJ>
int run_linearize(ddp_hip_ctx* ctx, const LinParams& p, uint32_t stages) {
  const Dims& d = ctx->d;
  const int64_t BT = d.batch * d.T;
  const int W = (int)(d.n + d.m);
  const int64_t P = (int64_t)W * (W - 1) / 2;
  const int fd_mode = ctx->model_h.fd_mode;
  const bool small = NJ <= 6;
  // large trees with analytic first order: the constraint chain runs on the analytic jacobians (lin_analytic.hip:
  // ana_eq_kernel), and in mode 1 its tensors come out of the same pass over the perturbed points as the dynamics' own
  const bool ana_large = !small && !ctx->model_h.first_order_fd && ctx->model_h.kind == DDP_HIP_MODEL_TREE;
  const bool eq_stage = (stages & DDP_HIP_LIN_EQ) && d.Etot > 0;
  const bool m1_fused = ana_large && fd_mode == 1 && p.has_tensors && eq_stage;   // LIN_SECOND's mode-1 pass is issued by the LIN_EQ stage
  if (stages & DDP_HIP_LIN_COST) hipLaunchKernelGGL(lin_cost_kernel, dim3((unsigned)BT), dim3(64), 0, ctx->stream, p);
  // static-topology path: the q- / v-caches of the mode-2 stencil also serve the first order (base configuration and
  // base (q, v)), so they are built ahead of whichever stage comes first
  bool caches_built = false;
  int static_rc = DDP_HIP_OK;
  auto build_caches = [&]() {
    if (caches_built || !p.qcache) return;
    const int nv = (int)d.nv;
    if (ctx->lin_static && getenv("DDP_HIP_NO_STATIC_CACHE") == nullptr) { const int rc_ = lin_static_launch(ctx, p, 5); if (rc_ != DDP_HIP_OK) static_rc = rc_; }
    else {
      hipLaunchKernelGGL((lin_qcache_kernel<NJ>), dim3(blocks_for(BT * (nv + 1))), dim3(LBS), 0, ctx->stream, p);
      hipLaunchKernelGGL((lin_vcache_kernel<NJ>), dim3(blocks_for(BT * (2 * nv + 1))), dim3(LBS), 0, ctx->stream, p);
    }
    caches_built = true;
  };
  if (stages & DDP_HIP_LIN_FIRST) {
    prof_begin(ctx, DDP_HIP_K_LIN_FIRST);
    hipLaunchKernelGGL((lin_base_kernel<NJ>), dim3(blocks_for(BT)), dim3(LBS), 0, ctx->stream, p);
    if (!ctx->model_h.first_order_fd && ctx->model_h.kind == DDP_HIP_MODEL_TREE) {
      if constexpr (small) hipLaunchKernelGGL((lin_first_analytic_small_kernel<NJ>), dim3(blocks_for(BT)), dim3(LBS), 0, ctx->stream, p);
      else {
        const bool m1_next = fd_mode == 1 && p.has_tensors && (stages & DDP_HIP_LIN_SECOND);
        const int rc_ = lin_analytic_launch(ctx, p, 0, LIN_ANA_F | (m1_next ? LIN_ANA_ACCEL : 0) | (m1_next && eq_stage ? LIN_ANA_EQ << 4 : 0));
        if (rc_ != DDP_HIP_OK) return rc_;
      }
    } else if (ctx->model_h.first_order_fd) {
      if (ctx->lin_static && p.qcache && getenv("DDP_HIP_NO_STATIC_FIRST") == nullptr) { build_caches(); { const int rc_ = lin_static_launch(ctx, p, 0); if (rc_ != DDP_HIP_OK) return rc_; } }
      else hipLaunchKernelGGL((lin_first_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p);
    }
    prof_end(ctx, DDP_HIP_K_LIN_FIRST);
  }
  if ((stages & DDP_HIP_LIN_SECOND) && p.has_tensors) {
    prof_begin(ctx, DDP_HIP_K_LIN_SECOND);
    if (fd_mode == 2) {
      if (p.qcache && p.ncfg > 1) {
        const int nv = (int)d.nv;
        const int64_t TRI = (int64_t)nv * (nv - 1) / 2, Pv = (int64_t)nv * nv + TRI, Pu = 2 * (int64_t)nv * nv + TRI;
        build_caches();
        if (ctx->lin_static && getenv("DDP_HIP_NO_STATIC_DIAG") == nullptr) { const int rc_ = lin_static_launch(ctx, p, 4); if (rc_ != DDP_HIP_OK) return rc_; }
        else hipLaunchKernelGGL((lin_diag_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p);
        // torque level first: on the static path its row kernel also forms the diagonal entries of the q and v directions
        if (ctx->lin_static) { const int rc_ = lin_static_launch(ctx, p, 3); if (rc_ != DDP_HIP_OK) return rc_; }
        else hipLaunchKernelGGL((lin_offdiag_kernel<NJ, 3>), dim3(blocks_for(BT * Pu)), dim3(LBS), 0, ctx->stream, p);
        if (ctx->lin_static) { const int rc_ = lin_static_launch(ctx, p, 2); if (rc_ != DDP_HIP_OK) return rc_; }
        else hipLaunchKernelGGL((lin_offdiag_kernel<NJ, 2>), dim3(blocks_for(BT * Pv)), dim3(LBS), 0, ctx->stream, p);
        if (ctx->lin_static && getenv("DDP_HIP_NO_STATIC_CFG") == nullptr) { const int rc_ = lin_static_launch(ctx, p, 1); if (rc_ != DDP_HIP_OK) return rc_; }
        else hipLaunchKernelGGL((lin_offdiag_kernel<NJ, 1>), dim3(blocks_for(BT * TRI)), dim3(LBS), 0, ctx->stream, p);
      } else {
        hipLaunchKernelGGL((lin_diag_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p);
        hipLaunchKernelGGL((lin_offdiag_kernel<NJ, 0>), dim3(blocks_for(BT * P)), dim3(LBS), 0, ctx->stream, p);
      }
    } else if (fd_mode == 1) {
      if (ctx->model_h.first_order_fd) return DDP_HIP_E_UNSUPPORTED;  // forward differences of FD jacobians are numerically void (refused at ddp_hip_create already)
      if constexpr (small) hipLaunchKernelGGL((second_m1_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p, 0);
      else if (!m1_fused) { const int rc_ = lin_analytic_launch(ctx, p, 1, LIN_ANA_F); if (rc_ != DDP_HIP_OK) return rc_; }
    } else {
      // fd_mode 0: Gauss-Newton variant, tensors are zero
      HIP_TRY(hipMemsetAsync(p.fxx, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_FXX].size * d.batch), ctx->stream));
      HIP_TRY(hipMemsetAsync(p.fux, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_FUX].size * d.batch), ctx->stream));
      HIP_TRY(hipMemsetAsync(p.fuu, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_FUU].size * d.batch), ctx->stream));
    }
    prof_end(ctx, DDP_HIP_K_LIN_SECOND);
  }
  if (eq_stage) {
    // small vector-space models with analytic jacobians (the pendulum, UR5-like arms in the drivers' mode) chain per lane; every
    // model with forward-differenced jacobians takes the three-kernel chain (round 3: the per-lane form differenced 2 x 18 full
    // dynamics evaluations and multiplied 12 x 12 matrices in ONE lane per (instance, t): 3.2 ms of latency at any size)
    bool chained = false;
    if constexpr (small) {
      if (!p.eq_xk) {
        chained = true;
        hipLaunchKernelGGL((eq_first_kernel<NJ>), dim3(blocks_for(BT)), dim3(LBS), 0, ctx->stream, p);
        if (p.has_tensors) {
          if (fd_mode == 2) {
            hipLaunchKernelGGL((eq_second_m2_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p, 0);
            hipLaunchKernelGGL((eq_second_m2_kernel<NJ>), dim3(blocks_for(BT * P)), dim3(LBS), 0, ctx->stream, p, 1);
          } else if (fd_mode == 1) {
            hipLaunchKernelGGL((second_m1_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p, 1);
          } else {
            HIP_TRY(hipMemsetAsync(p.eq_xx, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_XX].size * d.batch), ctx->stream));
            HIP_TRY(hipMemsetAsync(p.eq_ux, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_UX].size * d.batch), ctx->stream));
            HIP_TRY(hipMemsetAsync(p.eq_uu, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_UU].size * d.batch), ctx->stream));
          }
        }
      }
    }
    if (chained) {
    } else if (ana_large) {
      // analytic jacobians: base point from the resident f_x, f_u, then the tensors
      { const int rc_ = lin_analytic_launch(ctx, p, 0, LIN_ANA_EQ); if (rc_ != DDP_HIP_OK) return rc_; }
      if (p.has_tensors) {
        if (fd_mode == 2) {
          hipLaunchKernelGGL((eq_second_m2_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p, 0);
          hipLaunchKernelGGL((eq_second_m2_kernel<NJ>), dim3(blocks_for(BT * P)), dim3(LBS), 0, ctx->stream, p, 1);
        } else if (fd_mode == 1) {
          const int fl = LIN_ANA_EQ | ((stages & DDP_HIP_LIN_SECOND) ? LIN_ANA_F : 0);
          if (fl & LIN_ANA_F) prof_begin(ctx, DDP_HIP_K_LIN_SECOND);
          const int rc_ = lin_analytic_launch(ctx, p, 1, fl);
          if (fl & LIN_ANA_F) prof_end(ctx, DDP_HIP_K_LIN_SECOND);
          if (rc_ != DDP_HIP_OK) return rc_;
        } else {
          HIP_TRY(hipMemsetAsync(p.eq_xx, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_XX].size * d.batch), ctx->stream));
          HIP_TRY(hipMemsetAsync(p.eq_ux, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_UX].size * d.batch), ctx->stream));
          HIP_TRY(hipMemsetAsync(p.eq_uu, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_UU].size * d.batch), ctx->stream));
        }
      }
    } else {
      // large models, forward-differenced jacobians: chain rule as three kernels; mode-2 second order reuses the per-point kernel
      if (!p.eq_xk) return DDP_HIP_E_UNSUPPORTED;
      const int K = ctx->model_h.eq_advance;
      if (K < 1) return DDP_HIP_E_UNSUPPORTED;
      hipLaunchKernelGGL((eq_chain_kernel<NJ>), dim3(blocks_for(BT)), dim3(LBS), 0, ctx->stream, p);
      if (K > 1) hipLaunchKernelGGL((eq_fdjac_kernel<NJ>), dim3(blocks_for(BT * (K - 1) * d.n)), dim3(LBS), 0, ctx->stream, p);
      hipLaunchKernelGGL(eq_combine_kernel, dim3((unsigned)BT), dim3(256), sizeof(double) * (size_t)(2 * d.emax * d.n), ctx->stream, p);
      if (p.has_tensors) {
        if (fd_mode == 2) {
          hipLaunchKernelGGL((eq_second_m2_kernel<NJ>), dim3(blocks_for(BT * W)), dim3(LBS), 0, ctx->stream, p, 0);
          hipLaunchKernelGGL((eq_second_m2_kernel<NJ>), dim3(blocks_for(BT * P)), dim3(LBS), 0, ctx->stream, p, 1);
        } else if (fd_mode == 1) {
          return DDP_HIP_E_UNSUPPORTED;   // forward differences of FD jacobians: refused at ddp_hip_create
        } else {
          HIP_TRY(hipMemsetAsync(p.eq_xx, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_XX].size * d.batch), ctx->stream));
          HIP_TRY(hipMemsetAsync(p.eq_ux, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_UX].size * d.batch), ctx->stream));
          HIP_TRY(hipMemsetAsync(p.eq_uu, 0, sizeof(double) * (size_t)(ctx->seq[DDP_HIP_SEQ_EQ_UU].size * d.batch), ctx->stream));
        }
      }
    }
  }
  HIP_TRY(hipGetLastError());
  return static_rc;
}

}  // namespace

int lin_setup(ddp_hip_ctx* ctx) {
  // q-part cache of the mode-2 stencil (tree models with resident tensors only)
  const bool tree = ctx->model_h.kind == DDP_HIP_MODEL_TREE;
  const bool tensors = ctx->model_h.fd_mode == 2 && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS);
  const bool want = tree && tensors && !ctx->model_h.ff && getenv("DDP_HIP_NO_QCACHE") == nullptr;   // the caches index q by joint: 1-DoF trees
  const int topo = (tree && getenv("DDP_HIP_NO_STATIC") == nullptr) ? lin_static_supported(ctx->model_h) : 0;
  const Dims& d = ctx->d;
  if (want) { ctx->lin_ncfg = (int32_t)d.nv + 1; ctx->lin_nvcfg = 2 * (int32_t)d.nv + 1; }
  else if (topo && ctx->model_h.first_order_fd) { ctx->lin_ncfg = 1; ctx->lin_nvcfg = 1; }   // first order only: base q, base (q, v)
  else if (topo && tree && !ctx->model_h.ff && d.nv > 6 && ctx->model_h.fd_mode == 1 && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS) &&
           getenv("DDP_HIP_ANA_OWN_ABA") == nullptr) {
    // analytic mode 1: the accelerations of its 2 nv perturbed points come from the static first-order kernels (lin_analytic.hip)
    ctx->lin_ncfg = 1; ctx->lin_nvcfg = 1;
  }
  if (ctx->lin_ncfg) {
    ctx->lin_ws_bytes = sizeof(double) * (size_t)(d.batch * d.T * ((int64_t)ctx->lin_ncfg * d.nv * rbd::QC_STRIDE + (int64_t)ctx->lin_nvcfg * d.nv * rbd::VC_STRIDE));
    HIP_TRY(hipMalloc(&ctx->lin_ws, ctx->lin_ws_bytes));
    ctx->lin_static = topo;
  }
  if (ctx->lin_static) {
    const int64_t BT = ctx->d.batch * ctx->d.T;
    int64_t slice = 1024;
    if (const char* ev = getenv("DDP_HIP_QWS_BT")) { const int v = atoi(ev); if (v >= 16 && v <= 65536) slice = v; }   // tuning knob
    ctx->lin_qws_bt = BT < slice ? BT : slice;
    HIP_TRY(hipMalloc(&ctx->lin_qws, sizeof(double) * (size_t)(ctx->lin_qws_bt * lin_static_ws_per_bt(ctx->model_h))));
    if (ctx->lin_ncfg > 1) {     // the mode-2 stencil is resident: its configuration level runs slice-pipelined on two streams
      HIP_TRY(hipMalloc(&ctx->lin_qws2, sizeof(double) * (size_t)(ctx->lin_qws_bt * lin_static_ws_per_bt(ctx->model_h))));
      HIP_TRY(hipStreamCreateWithFlags(&ctx->lin_stream2, hipStreamNonBlocking));
      for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&ctx->lin_ev_up[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->lin_ev_dn[k], hipEventDisableTiming));
      }
    }
  }
  {
    // analytic first order on large trees: its own kernels and workspace (incl. the constraint chain on the analytic
    // jacobians, lin_analytic.hip: ana_eq_kernel).  Mode 1 on forward-differenced jacobians is numerically void -- eps_mach /
    // sqrt(eps_mach)^2 = O(1) noise -- and the reference cannot express it (its first order is always analytic): refused here
    if (ctx->model_h.fd_mode == 1 && ctx->model_h.first_order_fd && !(ctx->flags & DDP_HIP_FLAG_NO_TENSORS)) return DDP_HIP_E_UNSUPPORTED;
    const int rc_ = lin_analytic_setup(ctx);
    if (rc_ != DDP_HIP_OK) return rc_;
  }
  // look-ahead states / jacobians of the constraint chain on large models
  if (ctx->d.Etot > 0 && tree && ctx->model_h.first_order_fd) {
    const Dims& d = ctx->d;
    const int64_t K = ctx->model_h.eq_advance;
    const size_t words = (size_t)(d.batch * d.T * (K * d.nx + (K > 1 ? K - 1 : 0) * d.n * d.n + d.emax * d.n));
    HIP_TRY(hipMalloc(&ctx->eq_ws, sizeof(double) * words));
  }
  return DDP_HIP_OK;
}
void lin_teardown(ddp_hip_ctx* ctx) {
  lin_analytic_teardown(ctx);
  for (int k = 0; k < 2; ++k) {
    if (ctx->lin_ev_up[k]) (void)hipEventDestroy(ctx->lin_ev_up[k]);
    if (ctx->lin_ev_dn[k]) (void)hipEventDestroy(ctx->lin_ev_dn[k]);
  }
  if (ctx->lin_stream2) (void)hipStreamDestroy(ctx->lin_stream2);
  if (ctx->lin_qws2) (void)hipFree(ctx->lin_qws2);
  if (ctx->lin_qws) (void)hipFree(ctx->lin_qws);
  if (ctx->eq_ws) (void)hipFree(ctx->eq_ws);
  if (ctx->lin_ws) (void)hipFree(ctx->lin_ws);
}

extern "C" int ddp_hip_linearize_stages(ddp_hip_ctx* ctx, uint32_t stages) {
  if (!ctx) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  LinParams p = make_params(ctx);
  int rc;
  const int nv = (int)ctx->d.nv;
  if ((stages & DDP_HIP_LIN_SECOND) && p.has_tensors && p.skip_top && !ctx->tensor_tops_zero) {
    // the configuration rows the static stencil leaves alone must hold zeros: once per context, and again after somebody
    // else has written to the tensors (upload / fill / device_ptr)
    const int64_t BT = ctx->d.batch * ctx->d.T, n = ctx->d.n, m = ctx->d.m;
    hipLaunchKernelGGL(tensor_zero_top_kernel, dim3(8192), dim3(256), 0, ctx->stream, p.fxx, BT * n * n, (int)n, nv);
    hipLaunchKernelGGL(tensor_zero_top_kernel, dim3(8192), dim3(256), 0, ctx->stream, p.fux, BT * m * n, (int)n, nv);
    hipLaunchKernelGGL(tensor_zero_top_kernel, dim3(8192), dim3(256), 0, ctx->stream, p.fuu, BT * m * m, (int)n, nv);
    HIP_TRY(hipGetLastError());
    ctx->tensor_tops_zero = true;
  }
  if (nv <= 1) rc = run_linearize<1>(ctx, p, stages);
  else if (nv <= 6 && !ctx->model_h.ff) rc = run_linearize<6>(ctx, p, stages);   // (the one-lane constraint chain of small models is vector-space only)
  else if (nv <= 38) rc = run_linearize<38>(ctx, p, stages);
  else rc = run_linearize<64>(ctx, p, stages);
  ctx->ana_M0_fresh = false;
  ctx->ana_A_fresh = false;
  if (rc != DDP_HIP_OK) return rc;
  END_SYNC(ctx);
  // mode 2 writes one value to both (i, j, k) and (i, k, j) (problem.hpp:283-292), mode 0 leaves zeros: f_xx is symmetric bit
  // for bit and the backward sweep reads one of each pair of mirrored half-slabs (bwd_split.h); mode 1's forward differences
  // of jacobians are not
  if ((stages & DDP_HIP_LIN_SECOND) && p.has_tensors) {
    ctx->tensors_sym = ctx->model_h.fd_mode == 2 || ctx->model_h.fd_mode == 0;
    ctx->fxx_mirror_pending = p.skip_qv_mirror != 0;
    // (the run-time-tree kernels and mode 0 leave the same structure -- it is a property of the stencil's values, not of who
    // writes them -- but only the static mode-2 path has been held to it bit for bit: tests/test_round3_boundary.py)
    ctx->tensor_tops_sparse = p.skip_top != 0;
  }
  return DDP_HIP_OK;
}

// FXX complete for a reader that does not know about the skipped block (download, device_ptr, the run-time-shaped sweep)
int lin_materialize_fxx(ddp_hip_ctx* ctx) {
  if (!ctx->fxx_mirror_pending) return DDP_HIP_OK;
  double* fxx = ctx->seq[DDP_HIP_SEQ_FXX].ptr;
  double* fuu = ctx->seq[DDP_HIP_SEQ_FUU].ptr;
  if (fxx && fuu) {
    const int64_t BT = ctx->d.batch * ctx->d.T;
    hipLaunchKernelGGL(tensor_mirror_kernel, dim3(8192), dim3(256), 0, ctx->stream, fxx, BT, (int)ctx->d.n, (int)ctx->d.n);
    hipLaunchKernelGGL(tensor_mirror_kernel, dim3(4096), dim3(256), 0, ctx->stream, fuu, BT, (int)ctx->d.n, (int)ctx->d.m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  ctx->fxx_mirror_pending = false;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_linearize(ddp_hip_ctx* ctx) {
  return ddp_hip_linearize_stages(ctx, DDP_HIP_LIN_COST | DDP_HIP_LIN_FIRST | DDP_HIP_LIN_SECOND | DDP_HIP_LIN_EQ);
}
