// bwd.hip -- backward (Riccati-like) sweep of augmented-Lagrangian DDP on gfx950.
//
// Replaces ddp_solver_t::backward_pass<primal_dual_affine_multipliers> (include/ddp/ddp_bwd.ipp:9-155).
// The recursion is strictly sequential in t; what is parallel is (a) the independent instances of
// the batch and (b), inside one step, the columns of Q_xx / Q_ux / Q_uu.  Each step is two launches:
//
//   bwd_assemble  grid (jobs, batch): job = a block of x-columns or u-columns.  Streams the
//                 timestep-contiguous slabs f_xx(:,:,c), f_ux(:,:,c), f_uu(:,:,c) once from HBM with
//                 16-byte coalesced loads, contracts them with V_x through LDS (tensor.hpp:179-198),
//                 and adds the dense terms f^T V_xx f (+ multiplier terms) of ddp_bwd.ipp:61-87.
//                 This is the HBM-bound kernel (0.25 flop/byte).
//   bwd_gains     grid (batch): LLT of Q_uu + reg I (lower triangle only, fail <=> pivot <= 0,
//                 ddp_bwd.ipp:104-105), k/K solves (:134-136), V_x / V_xx update (:142-146), all in LDS.
//
// A non-positive pivot marks the instance failed; the reg/mu rule of ddp_bwd.ipp:106-110 is applied
// on the device and the host relaunches the sweep for the failed instances only.
#include <stdio.h>
#include <stdlib.h>

#include "internal.h"

namespace {

struct BwdParams {
  Dims d;
  const int64_t* ne;
  const int64_t* Epre;
  const double *lfx, *lfxx, *lx, *lu, *lxx, *lux, *luu, *fx, *fu, *fxx, *fux, *fuu;
  const double *eq_val, *eq_x, *eq_u, *eq_xx, *eq_ux, *eq_uu;
  const double *x, *mult_val, *mult_jac;
  double *fb_origin, *fb_val, *fb_jac, *vx_trace, *vxx_trace;
  double *ws_V, *ws_Q, *ws_D, *reg, *mu;
  int32_t* status;
  int64_t* restarts;
  const BwdJob* jobs;
  const BwdJob* jobs_half;   // K3h's job list (bwd_split.h: bwd_contract_half), or null
  int32_t njobs_half;
  int32_t half_mode;         // 1: symmetric tensors with two-entry upper halves (static mode-2 stencil); 2: this context's analytic mode-1
                             // tensors -- no symmetry, the upper half of every f_xx / f_ux column and all of f_uu exact zeros
  int32_t has_tensors;
  int32_t b0;          // first instance of the group this launch sweeps
  int32_t c_accumulate; // K3's epilogue adds its contraction to what is already in the Q workspace (bwd_v2.h) instead of storing it
  int32_t sym_tensors;  // f_xx is symmetric in its two input indices bit for bit (mode-2 / zero tensors of this context's own linearisation):
                        // K3 reads one of each pair of mirrored half-slabs (bwd_split.h, job kind 2)
};

constexpr int BS = 256;
typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(BS) void bwd_init(BwdParams p) {
  const int b = p.b0 + blockIdx.x;
  const int s = p.status[b];
  if (s == 2) return;
  const int64_t n = p.d.n;
  double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  double* Vxx = Vx + n;
  const double* lfx = p.lfx + (int64_t)b * n;
  const double* lfxx = p.lfxx + (int64_t)b * n * n;
  for (int64_t i = threadIdx.x; i < n; i += BS) Vx[i] = lfx[i];        // ddp_bwd.ipp:28
  for (int64_t i = threadIdx.x; i < n * n; i += BS) Vxx[i] = lfxx[i];  // ddp_bwd.ipp:27
  __syncthreads();
  if (threadIdx.x == 0 && s == 1) p.status[b] = 0;
}

// out[j] += sum_i v[i] * Tn[i + j*O]  for j < L   (tensor_view_t::noalias_contract_add_outdim,
// detail/tensor.hpp:179-198, for one (O x L) slab = one value of the right index).
// Tn is 16-byte aligned and O is even, so a lane's 16-byte load never straddles a column.
// Stage 1: every lane loads consecutive double2's (1 KiB per wave instruction) and writes its two-term
//          partial to LDS at [column][pair] with the column stride padded to O/2+1 (bank-conflict free).
// Stage 2: lane j sums the O/2 partials of column j in a fixed order (deterministic).
template <int OC>
__device__ __forceinline__ void contract_slab(const double* __restrict__ Tn, int O_rt, int L, const double* s_v,
                                              double* s_part, double* out, int out_stride) {
  const int O = OC > 0 ? OC : O_rt;
  const int hp = O >> 1;
  const int ld = hp + 1;
  const int total = hp * L;
  const f64x2* __restrict__ T2 = reinterpret_cast<const f64x2*>(Tn);
  const int tid = threadIdx.x;
#pragma unroll 4
  for (int f = tid; f < total; f += BS) {
    const f64x2 a = __builtin_nontemporal_load(&T2[f]);
    const int j = f / hp;
    const int ip = f - j * hp;
    s_part[j * ld + ip] = s_v[2 * ip] * a.x + s_v[2 * ip + 1] * a.y;
  }
  __syncthreads();
  for (int j = tid; j < L; j += BS) {
    double s = 0.0;
    const double* pj = s_part + j * ld;
    for (int k = 0; k < hp; ++k) s += pj[k];
    out[j * out_stride] += s;
  }
  __syncthreads();
}

template <int NC, int MC>
__global__ __launch_bounds__(BS) void bwd_assemble(BwdParams p, int64_t t) {
  const int b = blockIdx.y;
  if (p.status[b] != 0) return;
  const BwdJob job = p.jobs[blockIdx.x];
  const int n = NC > 0 ? NC : (int)p.d.n;
  const int m = MC > 0 ? MC : (int)p.d.m;
  const int64_t T = p.d.T;
  const int e = (int)p.ne[t];
  const int64_t Eo = p.Epre[t];
  const int64_t Etot = p.d.Etot;
  const double mu = p.mu[b];
  const int tid = threadIdx.x;
  const int kind = job.kind, c0 = job.c0, cn = job.cn;
  const int rows = kind == 0 ? n + m : m;

  const double* Vx = p.ws_V + (int64_t)b * (n + (int64_t)n * n);
  const double* Vxx = Vx + n;
  double* Q = p.ws_Q + (int64_t)b * (n + m + (int64_t)n * n + (int64_t)m * n + (int64_t)m * m);
  double* Qx = Q;
  double* Qu = Qx + n;
  double* Qxx = Qu + m;
  double* Qux = Qxx + (int64_t)n * n;
  double* Quu = Qux + (int64_t)m * n;

  const int64_t bt = (int64_t)b * T + t;
  const double* fx = p.fx + bt * n * n;
  const double* fu = p.fu + bt * n * m;
  const double* eqv = p.eq_val + (int64_t)b * Etot + Eo;
  const double* eqx = p.eq_x + ((int64_t)b * Etot + Eo) * n;
  const double* equ = p.eq_u + ((int64_t)b * Etot + Eo) * m;
  const double* pe = p.mult_val + (int64_t)b * Etot + Eo;
  const double* pex = p.mult_jac + ((int64_t)b * Etot + Eo) * n;

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_v = smem;                         // n
  double* s_tmp = s_v + n;                    // emax (>= e)
  double* s_W = s_tmp + p.d.emax;             // n * cn
  double* s_out = s_W + n * cn;               // rows * cn
  double* s_part = s_out + (n + m) * cn;      // (n/2+1) * max(n,m)

  for (int i = tid; i < n; i += BS) s_v[i] = Vx[i];
  for (int i = tid; i < e; i += BS) s_tmp[i] = pe[i] + mu * eqv[i];   // ddp_bwd.ipp:46
  __syncthreads();

  // W = V_xx * F(:, c0:c0+cn)  with F = f_x (x-job) or f_u (u-job)
  const double* Fc = (kind == 0 ? fx : fu) + (int64_t)c0 * n;
  for (int idx = tid; idx < n * cn; idx += BS) {
    const int r = idx % n, c = idx / n;
    double s = 0.0;
    for (int l = 0; l < n; ++l) s += Vxx[r + l * n] * Fc[l + c * n];
    s_W[idx] = s;
  }
  __syncthreads();

  // dense terms, in the reference's order: l, f^T V_xx f, then the multiplier terms
  for (int idx = tid; idx < rows * cn; idx += BS) {
    const int r = idx % rows, c = idx / rows;
    const int col = c0 + c;
    double acc;
    const double* Fr;
    if (kind == 0) {
      if (r < n) { acc = p.lxx[bt * n * n + r + (int64_t)col * n]; Fr = fx + (int64_t)r * n; }          // :70-71
      else { acc = p.lux[bt * m * n + (r - n) + (int64_t)col * m]; Fr = fu + (int64_t)(r - n) * n; }   // :83-84
    } else {
      acc = p.luu[bt * m * m + r + (int64_t)col * m]; Fr = fu + (int64_t)r * n;                         // :77-78
    }
    double s = 0.0;
    for (int l = 0; l < n; ++l) s += Fr[l] * s_W[l + c * n];
    acc += s;
    if (e > 0) {
      if (kind == 0) {
        if (r < n) {
          double s1 = 0.0, s2 = 0.0;
          for (int i = 0; i < e; ++i) {
            const double tmp2 = pex[i + col * e] + mu * eqx[i + col * e];   // :47
            s1 += eqx[i + r * e] * tmp2;                                     // :72
            s2 += pex[i + r * e] * eqx[i + col * e];                         // :73
          }
          acc += s1;
          acc += s2;
        } else {
          double s1 = 0.0;
          for (int i = 0; i < e; ++i) s1 += equ[i + (r - n) * e] * (pex[i + col * e] + mu * eqx[i + col * e]);  // :85
          acc += s1;
        }
      } else {
        double s1 = 0.0;
        for (int i = 0; i < e; ++i) s1 += equ[i + r * e] * equ[i + col * e];
        acc += s1 * mu;                                                      // :79
      }
      if (p.has_tensors) {
        // multiplier-weighted constraint tensors (:74, :80, :86); e is small, done in place
        const double* Te;
        int Ld;
        int jrow;
        if (kind == 0) {
          if (r < n) { Te = p.eq_xx + ((int64_t)b * Etot + Eo) * n * n; Ld = n; jrow = r; }
          else { Te = p.eq_ux + ((int64_t)b * Etot + Eo) * m * n; Ld = m; jrow = r - n; }
        } else { Te = p.eq_uu + ((int64_t)b * Etot + Eo) * m * m; Ld = m; jrow = r; }
        double s3 = 0.0;
        const double* col_ptr = Te + ((int64_t)jrow + (int64_t)col * Ld) * e;
        for (int i = 0; i < e; ++i) s3 += s_tmp[i] * col_ptr[i];
        acc += s3;
      }
    }
    s_out[idx] = acc;
  }
  // Q_x / Q_u entries of this job's columns (:61-68)
  for (int c = tid; c < cn; c += BS) {
    const int col = c0 + c;
    double acc, s = 0.0;
    if (kind == 0) {
      acc = p.lx[bt * n + col];
      for (int l = 0; l < n; ++l) s += fx[l + (int64_t)col * n] * s_v[l];
      acc += s;
      double s1 = 0.0, s2 = 0.0;
      for (int i = 0; i < e; ++i) { s1 += eqx[i + col * e] * s_tmp[i]; s2 += pex[i + col * e] * eqv[i]; }
      acc += s1;
      acc += s2;
      Qx[col] = acc;
    } else {
      acc = p.lu[bt * m + col];
      for (int l = 0; l < n; ++l) s += fu[l + (int64_t)col * n] * s_v[l];
      acc += s;
      double s1 = 0.0;
      for (int i = 0; i < e; ++i) s1 += equ[i + col * e] * s_tmp[i];
      acc += s1;
      Qu[col] = acc;
    }
  }
  __syncthreads();

  // second-order dynamics tensors contracted with V_x (:75, :81, :87): the HBM stream
  if (p.has_tensors) {
    for (int c = 0; c < cn; ++c) {
      const int col = c0 + c;
      if (kind == 0) {
        const double* Txx = p.fxx + (bt * n + col) * (int64_t)n * n;   // f_xx(:,:,col): n x n slab
        contract_slab<NC>(Txx, n, n, s_v, s_part, s_out + c * rows, 1);
        const double* Tux = p.fux + (bt * n + col) * (int64_t)n * m;   // f_ux(:,:,col): n x m slab
        contract_slab<NC>(Tux, n, m, s_v, s_part, s_out + c * rows + n, 1);
      } else {
        const double* Tuu = p.fuu + (bt * m + col) * (int64_t)n * m;   // f_uu(:,:,col): n x m slab
        contract_slab<NC>(Tuu, n, m, s_v, s_part, s_out + c * rows, 1);
      }
    }
  }

  for (int idx = tid; idx < rows * cn; idx += BS) {
    const int r = idx % rows, c = idx / rows;
    const int col = c0 + c;
    if (kind == 0) {
      if (r < n) Qxx[r + (int64_t)col * n] = s_out[idx];
      else Qux[(r - n) + (int64_t)col * m] = s_out[idx];
    } else {
      Quu[r + (int64_t)col * m] = s_out[idx];
    }
  }
}

template <int NC, int MC>
__global__ __launch_bounds__(BS) void bwd_gains(BwdParams p, int64_t t) {
  const int b = blockIdx.x;
  if (p.status[b] != 0) return;
  const int n = NC > 0 ? NC : (int)p.d.n;
  const int m = MC > 0 ? MC : (int)p.d.m;
  const int nx = (int)p.d.nx;
  const int64_t T = p.d.T;
  const int tid = threadIdx.x;
  const int64_t bt = (int64_t)b * T + t;

  double* Vx = p.ws_V + (int64_t)b * (n + (int64_t)n * n);
  double* Vxx = Vx + n;
  const double* Q = p.ws_Q + (int64_t)b * (n + m + (int64_t)n * n + (int64_t)m * n + (int64_t)m * m);
  const double* Qx = Q;
  const double* Qu = Qx + n;
  const double* Qxx = Qu + m;
  const double* Qux = Qxx + (int64_t)n * n;
  const double* Quu = Qux + (int64_t)m * n;

  const int lda = m | 1;   // odd leading dimensions: column walks by the lanes are bank-conflict free
  const int ldr = m | 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* A = smem;                    // lda * m      Q_uu + reg I, then its Cholesky factor (lower)
  double* R = A + lda * m;             // ldr * (n+1)  [-Q_u | -Q_ux] -> [k | K]
  double* S = R + ldr * (n + 1);       // ldr * n      Q_ux

  const double reg = p.reg[b];
  for (int idx = tid; idx < m * m; idx += BS) {
    const int i = idx % m, j = idx / m;
    A[i + j * lda] = Quu[idx] + (i == j ? reg : 0.0);                        // ddp_bwd.ipp:104
  }
  for (int i = tid; i < m; i += BS) R[i] = -Qu[i];                           // :135
  for (int idx = tid; idx < m * n; idx += BS) {
    const int i = idx % m, j = idx / m;
    const double q = Qux[idx];
    R[i + (j + 1) * ldr] = -q;                                               // :136
    S[i + j * ldr] = q;
  }
  __syncthreads();

  // Cholesky, lower triangle only; the update order per entry (k ascending) equals the left-looking
  // order of Eigen's unblocked LLT.  Failure <=> a pivot is <= 0.
  bool failed = false;
  for (int k = 0; k < m; ++k) {
    const double piv = A[k + k * lda];
    if (piv <= 0.0) { failed = true; break; }
    const double dk = sqrt(piv);
    for (int i = k + 1 + tid; i < m; i += BS) A[i + k * lda] = A[i + k * lda] / dk;
    __syncthreads();
    const int w = m - k - 1;
    for (int idx = tid; idx < w * w; idx += BS) {
      const int i = k + 1 + idx % w, j = k + 1 + idx / w;
      if (j <= i) A[i + j * lda] -= A[i + k * lda] * A[j + k * lda];
    }
    if (tid == 0) A[k + k * lda] = dk;
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      double r = p.reg[b], mu = p.mu[b];
      if (r < mu) r = mu;          // ddp_bwd.ipp:106-108
      mu *= 2;                     // :109
      r *= 2;                      // :110
      p.reg[b] = r;
      p.mu[b] = mu;
      p.status[b] = 1;
      p.restarts[b] += 1;
    }
    return;
  }

  // k = -(L L^T)^-1 Q_u,  K = -(L L^T)^-1 Q_ux: one right-hand side per lane
  for (int c = tid; c < n + 1; c += BS) {
    double* r = R + c * ldr;
    for (int i = 0; i < m; ++i) {
      double s = r[i];
      for (int l = 0; l < i; ++l) s -= A[i + l * lda] * r[l];
      r[i] = s / A[i + i * lda];
    }
    for (int i = m - 1; i >= 0; --i) {
      double s = r[i];
      for (int l = i + 1; l < m; ++l) s -= A[l + i * lda] * r[l];
      r[i] = s / A[i + i * lda];
    }
  }
  __syncthreads();

  double* fbo = p.fb_origin + bt * nx;
  double* fbk = p.fb_val + bt * m;
  double* fbK = p.fb_jac + bt * m * n;
  const double* xt = p.x + ((int64_t)b * (T + 1) + t) * nx;
  for (int i = tid; i < nx; i += BS) fbo[i] = xt[i];                         // :134
  for (int i = tid; i < m; i += BS) fbk[i] = R[i];
  for (int idx = tid; idx < m * n; idx += BS) fbK[idx] = R[idx % m + (idx / m + 1) * ldr];

  // V_x = Q_x + Q_ux^T k (:142-143);  V_xx = Q_xx + Q_ux^T K (:145-146)
  for (int i = tid; i < n; i += BS) {
    double s = 0.0;
    for (int l = 0; l < m; ++l) s += S[l + i * ldr] * R[l];
    const double v = Qx[i] + s;
    Vx[i] = v;
    if (p.vx_trace) p.vx_trace[bt * n + i] = v;
  }
  for (int idx = tid; idx < n * n; idx += BS) {
    const int i = idx % n, j = idx / n;
    double s = 0.0;
    const double* si = S + i * ldr;
    const double* kj = R + (j + 1) * ldr;
    for (int l = 0; l < m; ++l) s += si[l] * kj[l];
    const double v = Qxx[idx] + s;
    Vxx[idx] = v;
    if (p.vxx_trace) p.vxx_trace[bt * n * n + idx] = v;
  }
  if (t == 0 && tid == 0) p.status[b] = 2;                                   // :149-151
}

#include "bwd_fast.h"
#include "bwd_split.h"
#include "bwd_v2.h"

size_t assemble_lds_bytes(const ddp_hip_ctx* ctx, int cn_max) {
  const Dims& d = ctx->d;
  int64_t L = d.n > d.m ? d.n : d.m;
  int64_t words = d.n + d.emax + d.n * cn_max + (d.n + d.m) * cn_max + (d.n / 2 + 1) * L;
  return (size_t)words * sizeof(double);
}
size_t gains_lds_bytes(const ddp_hip_ctx* ctx) {
  const Dims& d = ctx->d;
  int64_t ld = d.m | 1;
  return (size_t)(ld * d.m + ld * (d.n + 1) + ld * d.n) * sizeof(double);
}

BwdParams make_params(ddp_hip_ctx* ctx) {
  BwdParams p{};
  p.d = ctx->d;
  p.ne = ctx->ne_d;
  p.Epre = ctx->Epre_d;
  auto S = [&](int s) { return ctx->seq[s].ptr; };
  p.lfx = S(DDP_HIP_SEQ_LFX); p.lfxx = S(DDP_HIP_SEQ_LFXX);
  p.lx = S(DDP_HIP_SEQ_LX); p.lu = S(DDP_HIP_SEQ_LU); p.lxx = S(DDP_HIP_SEQ_LXX); p.lux = S(DDP_HIP_SEQ_LUX); p.luu = S(DDP_HIP_SEQ_LUU);
  p.fx = S(DDP_HIP_SEQ_FX); p.fu = S(DDP_HIP_SEQ_FU);
  p.fxx = S(DDP_HIP_SEQ_FXX); p.fux = S(DDP_HIP_SEQ_FUX); p.fuu = S(DDP_HIP_SEQ_FUU);
  p.eq_val = S(DDP_HIP_SEQ_EQ_VAL); p.eq_x = S(DDP_HIP_SEQ_EQ_X); p.eq_u = S(DDP_HIP_SEQ_EQ_U);
  p.eq_xx = S(DDP_HIP_SEQ_EQ_XX); p.eq_ux = S(DDP_HIP_SEQ_EQ_UX); p.eq_uu = S(DDP_HIP_SEQ_EQ_UU);
  p.x = S(DDP_HIP_SEQ_X);
  p.mult_val = S(DDP_HIP_SEQ_MULT_VAL); p.mult_jac = S(DDP_HIP_SEQ_MULT_JAC);
  p.fb_origin = S(DDP_HIP_SEQ_FB_ORIGIN); p.fb_val = S(DDP_HIP_SEQ_FB_VAL); p.fb_jac = S(DDP_HIP_SEQ_FB_JAC);
  p.vx_trace = S(DDP_HIP_SEQ_VX_TRACE); p.vxx_trace = S(DDP_HIP_SEQ_VXX_TRACE);
  p.ws_V = ctx->ws_V; p.ws_Q = ctx->ws_Q; p.ws_D = ctx->ws_D; p.reg = ctx->reg_d; p.mu = ctx->mu_d;
  p.status = ctx->status_d; p.restarts = ctx->restarts_d;
  p.sym_tensors = (ctx->tensors_sym && ctx->jobs_sym_d && getenv("DDP_HIP_K3_NO_SYM") == nullptr) ? 1 : 0;
  p.jobs = p.sym_tensors ? ctx->jobs_sym_d : ctx->jobs_d;
  // K3h needs both structural facts: symmetry and the zero configuration rows (the static stencil's own tensors)
  const bool half = p.sym_tensors && ctx->tensor_tops_zero && ctx->tensor_tops_sparse && ctx->jobs_half_d && getenv("DDP_HIP_K3_NO_HALF") == nullptr;
  // ... or the structure the analytic mode-1 pass leaves (lin_analytic.hip): q+ = q + dt v has constant jacobian rows and M^-1 does
  // not depend on u, so the upper halves and f_uu are zeros it wrote itself
  const bool half_m1 = !half && ctx->fuu_zero && ctx->model_h.fd_mode == 1 && ctx->jobs_half_d && getenv("DDP_HIP_K3_NO_HALF") == nullptr;
  p.jobs_half = (half || half_m1) ? ctx->jobs_half_d : nullptr;
  p.njobs_half = (half || half_m1) ? ctx->njobs_half : 0;
  p.half_mode = half ? 1 : (half_m1 ? 2 : 0);
  p.has_tensors = (ctx->flags & DDP_HIP_FLAG_NO_TENSORS) ? 0 : 1;
  return p;
}

template <int NC, int MC>
int launch_sweep(ddp_hip_ctx* ctx, const BwdParams& p, size_t lds_a, size_t lds_g) {
  const Dims& d = ctx->d;
  hipLaunchKernelGGL(bwd_init, dim3((unsigned)d.batch), dim3(BS), 0, ctx->stream, p);
  for (int64_t t = d.T - 1; t >= 0; --t) {
    prof_begin(ctx, DDP_HIP_K_BWD_ASSEMBLE);
    hipLaunchKernelGGL((bwd_assemble<NC, MC>), dim3((unsigned)ctx->njobs, (unsigned)d.batch), dim3(BS), lds_a, ctx->stream, p, t);
    prof_end(ctx, DDP_HIP_K_BWD_ASSEMBLE);
    prof_begin(ctx, DDP_HIP_K_BWD_GAINS);
    hipLaunchKernelGGL((bwd_gains<NC, MC>), dim3((unsigned)d.batch), dim3(BS), lds_g, ctx->stream, p, t);
    prof_end(ctx, DDP_HIP_K_BWD_GAINS);
  }
  HIP_TRY(hipGetLastError());
  return DDP_HIP_OK;
}

// split path (compile-time shapes): K3 bwd_contract streams the tensors, K4 bwd_riccati does the rest.
// The recursion chains K4(t+1) -> K3(t) -> K4(t) per instance, and K4 is a latency chain on one workgroup per instance
// (a whole CU's LDS each): run back to back on one stream, K3 leaves HBM idle for as long as K4 takes.  The batch is
// therefore swept in groups on their own streams, so that K3 of one group streams while K4 of the others factorises.
template <int NC, int MC>
int launch_sweep_split(ddp_hip_ctx* ctx, const BwdParams& p0) {
  const Dims& d = ctx->d;
  const int cn_max = ctx->cbx > ctx->cbu ? ctx->cbx : ctx->cbu;
  const size_t lds_c = sizeof(double) * (size_t)(NC + (NC + MC) * cn_max + 2 * (NC / 2 + 1) * MC);
  const size_t lds_h = sizeof(double) * (size_t)(NC + 3 * NC + 2 * (MC / 2 + 1) * NC);
  const size_t lds_r = sizeof(double) * (size_t)(2 * NC * (NC + MC));
  const int G = ctx->bwd_groups;
  const int64_t per = (d.batch + G - 1) / G;
  if (G > 1) {
    HIP_TRY(hipEventRecord(ctx->bwd_ev_start, ctx->stream));
    for (int g = 0; g < G; ++g) HIP_TRY(hipStreamWaitEvent(ctx->bwd_stream[g], ctx->bwd_ev_start, 0));
  }
  BwdParams pg[8];
  unsigned nb[8];
  hipStream_t st[8];
  for (int g = 0; g < G; ++g) {
    pg[g] = p0;
    pg[g].b0 = (int32_t)(g * per);
    const int64_t n_ = d.batch - g * per < per ? d.batch - g * per : per;
    nb[g] = n_ > 0 ? (unsigned)n_ : 0u;
    st[g] = G > 1 ? ctx->bwd_stream[g] : ctx->stream;
    if (!nb[g]) continue;
    hipLaunchKernelGGL(bwd_init, dim3(nb[g]), dim3(BS), 0, st[g], pg[g]);
    hipLaunchKernelGGL((bwd_dense0<NC, MC>), dim3(nb[g]), dim3(BSR), lds_r, st[g], pg[g]);
  }
  for (int64_t t = d.T - 1; t >= 0; --t) {
    for (int g = 0; g < G; ++g) {
      if (!nb[g]) continue;
      if (p0.has_tensors) {
        prof_begin(ctx, DDP_HIP_K_BWD_ASSEMBLE, st[g]);
        if (pg[g].jobs_half) hipLaunchKernelGGL((bwd_contract_half<NC, MC>), dim3((unsigned)pg[g].njobs_half, nb[g]), dim3(BSF), lds_h, st[g], pg[g], t);
        else hipLaunchKernelGGL((bwd_contract<NC, MC>), dim3((unsigned)ctx->njobs, nb[g]), dim3(BSF), lds_c, st[g], pg[g], t);
        prof_end(ctx, DDP_HIP_K_BWD_ASSEMBLE, st[g]);
      }
      prof_begin(ctx, DDP_HIP_K_BWD_GAINS, st[g]);
      hipLaunchKernelGGL((bwd_riccati<NC, MC>), dim3(nb[g]), dim3(BSR), lds_r, st[g], pg[g], t);
      prof_end(ctx, DDP_HIP_K_BWD_GAINS, st[g]);
    }
  }
  HIP_TRY(hipGetLastError());
  if (G > 1)
    for (int g = 0; g < G; ++g) {
      HIP_TRY(hipEventRecord(ctx->bwd_ev_done[g], ctx->bwd_stream[g]));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->bwd_ev_done[g], 0));
    }
  return DDP_HIP_OK;
}

// three-kernel step (bwd_v2.h): K5 dense terms on 8 workgroups per instance -> K3 tensor stream (+ P) -> K4' lean Riccati
template <int NC, int MC>
int enqueue_sweep_v2(ddp_hip_ctx* ctx, const BwdParams& p0) {
  const Dims& d = ctx->d;
  const int cn_max = ctx->cbx > ctx->cbu ? ctx->cbx : ctx->cbu;
  size_t lds_c = sizeof(double) * (size_t)(NC + (NC + MC) * cn_max + 2 * (NC / 2 + 1) * MC);
  if (ctx->bwd_k3_lds_pad > lds_c) lds_c = ctx->bwd_k3_lds_pad;
  const size_t lds_h = sizeof(double) * (size_t)(NC + 3 * NC + 2 * (MC / 2 + 1) * NC);   // development: limits K3 to one workgroup per CU (room for K4' / K5 of another group)
  const size_t lds_5 = sizeof(double) * (size_t)(NC * NC + NC * (NC + MC) + NC + d.emax);
  const unsigned nblk5 = (unsigned)((NC + MC + CB5 - 1) / CB5);
  const int G = ctx->bwd_groups;
  // K5 (dense terms) and K3 (tensor stream) of a step both depend on K4' of the step before and on nothing else: with tensors,
  // K5 goes to a side stream and K3 keeps its contracted blocks in a workspace of its own; K4' waits for both and forms P + C
  const bool fork = ctx->bwd_fork && G == 1 && p0.has_tensors;
  const int64_t per = (d.batch + G - 1) / G;
  if (G > 1) {
    HIP_TRY(hipEventRecord(ctx->bwd_ev_start, ctx->stream));
    for (int g = 0; g < G; ++g) HIP_TRY(hipStreamWaitEvent(ctx->bwd_stream[g], ctx->bwd_ev_start, 0));
  }
  BwdParams pg[8];
  unsigned nb[8];
  hipStream_t st[8];
  for (int g = 0; g < G; ++g) {
    pg[g] = p0;
    pg[g].b0 = (int32_t)(g * per);
    pg[g].c_accumulate = fork ? 2 : 1;
    const int64_t n_ = d.batch - g * per < per ? d.batch - g * per : per;
    nb[g] = n_ > 0 ? (unsigned)n_ : 0u;
    st[g] = G > 1 ? ctx->bwd_stream[g] : ctx->stream;
    if (nb[g]) hipLaunchKernelGGL(bwd_init, dim3(nb[g]), dim3(BS), 0, st[g], pg[g]);
  }
  for (int64_t t = d.T - 1; t >= 0; --t) {
    for (int g = 0; g < G; ++g) {
      if (!nb[g]) continue;
      if (fork) {
        HIP_TRY(hipEventRecord(ctx->bwd_ev_fork, st[g]));
        HIP_TRY(hipStreamWaitEvent(ctx->bwd_side, ctx->bwd_ev_fork, 0));
        hipLaunchKernelGGL((bwd_dense2<NC, MC>), dim3(nblk5, nb[g]), dim3(BS5), lds_5, ctx->bwd_side, pg[g], t);
        HIP_TRY(hipEventRecord(ctx->bwd_ev_join, ctx->bwd_side));
      } else {
        prof_begin(ctx, DDP_HIP_K_BWD_GAINS, st[g]);
        hipLaunchKernelGGL((bwd_dense2<NC, MC>), dim3(nblk5, nb[g]), dim3(BS5), lds_5, st[g], pg[g], t);
        prof_end(ctx, DDP_HIP_K_BWD_GAINS, st[g]);
      }
      if (p0.has_tensors) {
        prof_begin(ctx, DDP_HIP_K_BWD_ASSEMBLE, st[g]);
        if (pg[g].jobs_half) hipLaunchKernelGGL((bwd_contract_half<NC, MC>), dim3((unsigned)pg[g].njobs_half, nb[g]), dim3(BSF), lds_h, st[g], pg[g], t);
        else hipLaunchKernelGGL((bwd_contract<NC, MC>), dim3((unsigned)ctx->njobs, nb[g]), dim3(BSF), lds_c, st[g], pg[g], t);
        prof_end(ctx, DDP_HIP_K_BWD_ASSEMBLE, st[g]);
      }
      if (fork) HIP_TRY(hipStreamWaitEvent(st[g], ctx->bwd_ev_join, 0));
      prof_begin(ctx, DDP_HIP_K_BWD_GAINS, st[g]);
      hipLaunchKernelGGL((bwd_gains2<NC, MC>), dim3(nb[g]), dim3(BS4), 0, st[g], pg[g], t);
      prof_end(ctx, DDP_HIP_K_BWD_GAINS, st[g]);
    }
  }
  HIP_TRY(hipGetLastError());
  if (G > 1)
    for (int g = 0; g < G; ++g) {
      HIP_TRY(hipEventRecord(ctx->bwd_ev_done[g], ctx->bwd_stream[g]));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->bwd_ev_done[g], 0));
    }
  return DDP_HIP_OK;
}

// The same enqueue, captured once into a hipGraph per state of the kernel arguments and replayed: the fork / join of
// the group streams becomes graph edges, the 600 launches per group one submission.  Profiled sweeps (HIP events around
// kernels) and DDP_HIP_BWD_NO_GRAPH=1 take the direct path.
template <int NC, int MC>
int build_sweep_graph(ddp_hip_ctx* ctx, const BwdParams& p0, uint64_t key_misc, ddp_hip_ctx::BwdGraph** out) {
  ddp_hip_ctx::BwdGraph* slot = &ctx->bwd_graph[ctx->bwd_graph_next];
  ctx->bwd_graph_next = (ctx->bwd_graph_next + 1) % 4;
  if (slot->exec) { (void)hipGraphExecDestroy(slot->exec); slot->exec = nullptr; }
  if (slot->graph) { (void)hipGraphDestroy(slot->graph); slot->graph = nullptr; }
  HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  const int rc = enqueue_sweep_v2<NC, MC>(ctx, p0);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
  if (rc != DDP_HIP_OK || e != hipSuccess || !graph) { (void)hipGetLastError(); if (graph) (void)hipGraphDestroy(graph); return rc != DDP_HIP_OK ? rc : DDP_HIP_E_HIP; }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipGraphDestroy(graph); return DDP_HIP_E_HIP; }
  slot->graph = graph; slot->exec = exec; slot->key_x = p0.x; slot->key_misc = key_misc;
  *out = slot;
  return DDP_HIP_OK;
}

template <int NC, int MC>
int launch_sweep_v2(ddp_hip_ctx* ctx, const BwdParams& p0) {
  const uint32_t bwd_mask = (2u << DDP_HIP_K_BWD_ASSEMBLE) | (2u << DDP_HIP_K_BWD_GAINS);
  // profiled sweeps take the direct path: events recorded by a graph's event-record nodes cannot be read back with
  // hipEventElapsedTime on this ROCm (hipErrorInvalidHandle -- tried)
  if (!ctx->bwd_use_graph || (ctx->profile_mask & bwd_mask)) return enqueue_sweep_v2<NC, MC>(ctx, p0);
  const uint64_t key_misc = (uint64_t)p0.has_tensors | ((uint64_t)(p0.vx_trace != nullptr) << 1) | ((uint64_t)p0.sym_tensors << 2) | ((uint64_t)(p0.jobs_half != nullptr) << 3) | ((uint64_t)p0.half_mode << 4);
  auto find = [&](const void* key_x) -> ddp_hip_ctx::BwdGraph* {
    for (auto& g : ctx->bwd_graph)
      if (g.exec && g.key_x == key_x && g.key_misc == key_misc) return &g;
    return nullptr;
  };
  ddp_hip_ctx::BwdGraph* slot = find(p0.x);
  if (!slot) {
    int rc = build_sweep_graph<NC, MC>(ctx, p0, key_misc, &slot);
    if (rc != DDP_HIP_OK) return rc;
    // the trajectory buffers trade places after every iteration (ddp_hip_swap_traj) and x is the one kernel argument that
    // follows them: the twin graph is built now as well, so that no later sweep pays for a capture
    const double* other = ctx->seq[DDP_HIP_SEQ_X_NEW].ptr;
    if (other && other != p0.x && !find(other)) {
      BwdParams p1 = p0;
      p1.x = other;
      ddp_hip_ctx::BwdGraph* twin = nullptr;
      rc = build_sweep_graph<NC, MC>(ctx, p1, key_misc, &twin);
      if (rc != DDP_HIP_OK) return rc;
      slot = find(p0.x);
      if (!slot) return DDP_HIP_E_HIP;
    }
  }
  HIP_TRY(hipGraphLaunch(slot->exec, ctx->stream));
  return DDP_HIP_OK;
}

}  // namespace

int bwd_setup(ddp_hip_ctx* ctx) {
  const Dims& d = ctx->d;
  const int64_t n = d.n, m = d.m, B = d.batch;
  HIP_TRY(hipMalloc(&ctx->ws_V, sizeof(double) * (size_t)(B * (n + n * n))));
  HIP_TRY(hipMalloc(&ctx->ws_Q, sizeof(double) * (size_t)(B * (n + m + n * n + m * n + m * m))));
  HIP_TRY(hipMalloc(&ctx->ws_D, sizeof(double) * (size_t)(B * (n * n + m * n + m * m))));
  HIP_TRY(hipMalloc(&ctx->reg_d, sizeof(double) * (size_t)B));
  HIP_TRY(hipMalloc(&ctx->mu_d, sizeof(double) * (size_t)B));
  HIP_TRY(hipMalloc(&ctx->status_d, sizeof(int32_t) * (size_t)B));
  HIP_TRY(hipMalloc(&ctx->restarts_d, sizeof(int64_t) * (size_t)B));

  // column jobs: one x-column streams 3 units (n*n + n*m doubles), one u-column 1 unit (n*m doubles);
  // aim at >= ~2048 workgroups per launch so that all 256 CUs hold several streaming workgroups
  int64_t units = 3 * n + m;
  int64_t want = 2048 / B;
  if (want < 1) want = 1;
  if (want > n + m) want = n + m;
  int64_t upj = (units + want - 1) / want;
  int64_t cbx = upj / 3; if (cbx < 1) cbx = 1; if (cbx > 8) cbx = 8;
  int64_t cbu = upj;     if (cbu < 1) cbu = 1; if (cbu > 16) cbu = 16;
  // Talos shape (K3): one x-column per job (3 units: two half-slabs of f_xx and one slab of f_ux) and three u-columns per job
  // (3 units of f_uu): 89 equal jobs of 69 KB per instance.  Measured at 64 instances: 66 us per launch (6.1 TB/s) against 74 us
  // (5.5 TB/s) for the 31 jobs of 3 / 9 columns the rule above picks.
  if (n == 76 && m == 38) { cbx = 1; cbu = 3; }
  if (const char* ev = getenv("DDP_HIP_BWD_CBX")) { int v = atoi(ev); if (v >= 1 && v <= 8) cbx = v; }    // tuning knobs
  if (const char* ev = getenv("DDP_HIP_BWD_CBU")) { int v = atoi(ev); if (v >= 1 && v <= 16) cbu = v; }
  ctx->cbx = (int32_t)cbx; ctx->cbu = (int32_t)cbu;
  std::vector<BwdJob> jobs;
  for (int64_t c = 0; c < n; c += cbx) jobs.push_back(BwdJob{0, (int32_t)c, (int32_t)((n - c) < cbx ? (n - c) : cbx), 0});
  for (int64_t c = 0; c < m; c += cbu) jobs.push_back(BwdJob{1, (int32_t)c, (int32_t)((m - c) < cbu ? (m - c) : cbu), 0});
  ctx->njobs = (int32_t)jobs.size();
  if (n == 76 && m == 38) {
    // K4 needs more than the default 64 KB of dynamic LDS.  The attribute is per device: it is set here, with the
    // context's device current, by every context (not behind a process-wide flag)
    const size_t lds_r = sizeof(double) * (size_t)(2 * 76 * (76 + 38));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_riccati<76, 38>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_dense0<76, 38>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
    const size_t lds_5 = sizeof(double) * (size_t)(76 * 76 + 76 * (76 + 38) + 76 + d.emax);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_dense2<76, 38>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_5));
  }
  if (n == 76 && m == 38) {
    // groups of the sweep on their own streams (tuning knob DDP_HIP_BWD_GROUPS).  Default 1: measured at 64 instances, two
    // and four groups do not overlap K3 with the other groups' K4' / K5 (K3's workgroups hold every VGPR of the CUs they
    // run on; 28.7 / 29.7 ms against 30.7 ms)
    int64_t G = 1;
    if (const char* ev = getenv("DDP_HIP_BWD_GROUPS")) { const int v = atoi(ev); if (v >= 1 && v <= 8) G = v; }
    ctx->bwd_use_graph = getenv("DDP_HIP_BWD_NO_GRAPH") ? 0 : 1;
    if (const char* ev = getenv("DDP_HIP_K3_LDS")) {
      ctx->bwd_k3_lds_pad = (size_t)atoi(ev) * 1024;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_contract<76, 38>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (G > B) G = B;
    ctx->bwd_groups = (int32_t)G;
    // DDP_HIP_BWD_FORK=1 (development): K5 on a side stream beside K3.  Measured at 64 seeds: 29.2 ms against 26.6 ms for the
    // serial chain -- with a K5 workgroup on every CU (116 KB of LDS, a quarter of the wave slots) K3 drops from 74 to 80 us
    // per launch and the fork / join edges cost more than the 17 us of K5 they hide.  Off by default.
    ctx->bwd_fork = getenv("DDP_HIP_BWD_FORK") ? 1 : 0;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->bwd_side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ctx->bwd_ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->bwd_ev_join, hipEventDisableTiming));
    if (G > 1) {
      HIP_TRY(hipEventCreateWithFlags(&ctx->bwd_ev_start, hipEventDisableTiming));
      for (int g = 0; g < G; ++g) {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->bwd_stream[g], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ctx->bwd_ev_done[g], hipEventDisableTiming));
      }
    }
  }
  HIP_TRY(hipMalloc(&ctx->jobs_d, sizeof(BwdJob) * jobs.size()));
  HIP_TRY(hipMemcpy(ctx->jobs_d, jobs.data(), sizeof(BwdJob) * jobs.size(), hipMemcpyHostToDevice));
  if (n == 76 && m == 38 && cbx == 1) {
    // the same list for symmetric tensors: the x-columns c >= m skip their first half-slab (job kind 2, bwd_split.h)
    std::vector<BwdJob> js = jobs;
    for (auto& j : js) if (j.kind == 0 && j.c0 >= m) j.kind = 2;
    HIP_TRY(hipMalloc(&ctx->jobs_sym_d, sizeof(BwdJob) * js.size()));
    HIP_TRY(hipMemcpy(ctx->jobs_sym_d, js.data(), sizeof(BwdJob) * js.size(), hipMemcpyHostToDevice));
    // K3h (bwd_contract_half): x-columns in pairs (3 units each), u-columns in groups of 6 (3 units), the rest in one
    std::vector<BwdJob> jh;
    for (int64_t c = 0; c < n; c += 2) jh.push_back(BwdJob{10, (int32_t)c, 2, 0});
    for (int64_t c = 0; c < m; c += 6) jh.push_back(BwdJob{11, (int32_t)c, (int32_t)((m - c) < 6 ? (m - c) : 6), 0});
    ctx->njobs_half = (int32_t)jh.size();
    HIP_TRY(hipMalloc(&ctx->jobs_half_d, sizeof(BwdJob) * jh.size()));
    HIP_TRY(hipMemcpy(ctx->jobs_half_d, jh.data(), sizeof(BwdJob) * jh.size(), hipMemcpyHostToDevice));
  }
  return DDP_HIP_OK;
}

void bwd_teardown(ddp_hip_ctx* ctx) {
  for (auto& g : ctx->bwd_graph) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  for (int g = 0; g < 8; ++g) {
    if (ctx->bwd_stream[g]) { (void)hipStreamSynchronize(ctx->bwd_stream[g]); (void)hipStreamDestroy(ctx->bwd_stream[g]); }
    if (ctx->bwd_ev_done[g]) (void)hipEventDestroy(ctx->bwd_ev_done[g]);
  }
  if (ctx->bwd_ev_start) (void)hipEventDestroy(ctx->bwd_ev_start);
  if (ctx->bwd_side) { (void)hipStreamSynchronize(ctx->bwd_side); (void)hipStreamDestroy(ctx->bwd_side); }
  if (ctx->bwd_ev_fork) (void)hipEventDestroy(ctx->bwd_ev_fork);
  if (ctx->bwd_ev_join) (void)hipEventDestroy(ctx->bwd_ev_join);
  if (ctx->ws_V) (void)hipFree(ctx->ws_V);
  if (ctx->ws_Q) (void)hipFree(ctx->ws_Q);
  if (ctx->ws_D) (void)hipFree(ctx->ws_D);
  if (ctx->reg_d) (void)hipFree(ctx->reg_d);
  if (ctx->mu_d) (void)hipFree(ctx->mu_d);
  if (ctx->status_d) (void)hipFree(ctx->status_d);
  if (ctx->restarts_d) (void)hipFree(ctx->restarts_d);
  if (ctx->jobs_d) (void)hipFree(ctx->jobs_d);
  if (ctx->jobs_sym_d) (void)hipFree(ctx->jobs_sym_d);
  if (ctx->jobs_half_d) (void)hipFree(ctx->jobs_half_d);
}

extern "C" int ddp_hip_backward(ddp_hip_ctx* ctx, double* reg_io, double* mu_io, int64_t* restarts_out, int64_t max_restarts) {
  if (!ctx || !reg_io || !mu_io) return DDP_HIP_E_ARG;
  const Dims& d = ctx->d;
  const int64_t B = d.batch;
  HIP_TRY(hipSetDevice(ctx->device));
  BwdParams p = make_params(ctx);
  if (p.has_tensors && (!p.fxx || !p.fux || !p.fuu)) return DDP_HIP_E_UNSUPPORTED;
  {
    // the static stencil leaves the f_xx block out that the symmetric sweep never reads: any other sweep needs it
    const bool fast = d.n == 76 && d.m == 38 && d.emax <= 52 && getenv("DDP_HIP_GENERIC_BWD") == nullptr;
    if (p.has_tensors && ctx->fxx_mirror_pending && !(fast && p.sym_tensors)) { const int rc_ = lin_materialize_fxx(ctx); if (rc_ != DDP_HIP_OK) return rc_; }
  }

  HIP_TRY(hipMemcpyAsync(ctx->reg_d, reg_io, sizeof(double) * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->mu_d, mu_io, sizeof(double) * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
  std::vector<int32_t> status((size_t)B);
  for (int64_t b = 0; b < B; ++b) status[(size_t)b] = ctx->active_h[(size_t)b] ? 0 : 2;   // a frozen instance counts as done
  HIP_TRY(hipMemcpyAsync(ctx->status_d, status.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->restarts_d, 0, sizeof(int64_t) * (size_t)B, ctx->stream));

  const int cn_max = ctx->cbx > ctx->cbu ? ctx->cbx : ctx->cbu;
  const size_t lds_a = assemble_lds_bytes(ctx, cn_max);
  const size_t lds_g = gains_lds_bytes(ctx);
  if (lds_a > 160 * 1024 || lds_g > 160 * 1024) return DDP_HIP_E_UNSUPPORTED;

  bool any_restart = false;
  int rc = DDP_HIP_OK;
  for (int64_t attempt = 0;; ++attempt) {
    // the Talos-like shape runs the split K3 / K4 kernels; every other shape (and DDP_HIP_GENERIC_BWD=1, kept for
    // cross-checking the two implementations against each other) the run-time-shaped pair
    const bool generic = getenv("DDP_HIP_GENERIC_BWD") != nullptr;
    // (K5 stages the multiplier jacobians over its V / F region: up to 52 constraint rows per step fit)
    if (d.n == 76 && d.m == 38 && d.emax <= 52 && !generic) rc = getenv("DDP_HIP_BWD_V1") ? launch_sweep_split<76, 38>(ctx, p) : launch_sweep_v2<76, 38>(ctx, p);
    else if (d.n == 12 && d.m == 6) rc = launch_sweep<12, 6>(ctx, p, lds_a, lds_g);
    else rc = launch_sweep<0, 0>(ctx, p, lds_a, lds_g);
    if (rc != DDP_HIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(status.data(), ctx->status_d, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    bool failed = false;
    for (int64_t b = 0; b < B; ++b) failed |= status[(size_t)b] == 1;
    if (!failed) break;
    any_restart = true;
    if (max_restarts >= 0 && attempt >= max_restarts) { rc = DDP_HIP_E_MAX_RESTARTS; break; }
  }
  HIP_TRY(hipMemcpyAsync(reg_io, ctx->reg_d, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(mu_io, ctx->mu_d, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
  if (restarts_out)
    HIP_TRY(hipMemcpyAsync(restarts_out, ctx->restarts_d, sizeof(int64_t) * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (rc != DDP_HIP_OK) return rc;
  return any_restart ? DDP_HIP_EV_LLT_RESTART : DDP_HIP_OK;
}

#ifdef BWD_STAMPS
extern "C" int ddp_hip_debug_stamps(unsigned long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_bwd_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
}
#endif
