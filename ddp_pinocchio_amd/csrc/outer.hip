// outer.hip -- device side of the outer augmented-Lagrangian loop of solve<M> (ddp.hpp:745-842): the pieces of
// update_derivatives (ddp.hpp:642-696) that sit between compute_derivatives and backward_pass, so that an iteration
// needs no derivative data on the host:
//   affine_vector_function_seq_t::update_origin   mat_seq_common.hpp:62-89   ddp_hip_update_origin
//   optimality_constr / optimality_obj             ddp.hpp:516-523, 576-627   ddp_hip_optimality
//   multiplier update p.val += mu (eq + eq_u k), p.jac += mu (eq_x + eq_u K)   ddp.hpp:680-688   ddp_hip_update_multipliers
// All O(T n^2) per instance: one workgroup per (instance, t) where the steps are independent, one per instance for the
// adjoint recursion of optimality_obj (sequential in t).
#include <math.h>

#include <vector>

#include "internal.h"
#include "lie.h"

namespace {

struct OuterParams {
  Dims d;
  const DevModel* model;
  const int64_t* ne;
  const int64_t* Epre;
  const double *x, *lfx, *lx, *lu, *fx, *fu, *eq_val, *eq_x, *eq_u;
  double *m_origin, *m_val, *m_jac;   // multipliers (rows ne[t])
  double *f_origin, *f_val, *f_jac;   // control feedback (rows m)
  const double* mu;                   // [batch], device
  double* out;                        // [batch][2]: optimality_obj, optimality_constr
};

constexpr int OBS = 128;

// val += jac (x_new - origin); origin = x_new (on a vector space d(x_new - origin)/dx_new = I leaves jac as it is,
// problem.hpp:414-439)
__global__ __launch_bounds__(OBS) void update_origin_kernel(OuterParams p, int which) {
  const int64_t t = blockIdx.x, T = p.d.T;
  const int b = blockIdx.y;
  const int n = (int)p.d.n, nx = (int)p.d.nx, m = (int)p.d.m;
  const int r = which == 0 ? (int)p.ne[t] : m;
  const int64_t R = which == 0 ? (int64_t)b * p.d.Etot + p.Epre[t] : ((int64_t)b * T + t) * m;
  double* org = (which == 0 ? p.m_origin : p.f_origin) + ((int64_t)b * T + t) * nx;
  double* val = (which == 0 ? p.m_val : p.f_val) + R;
  const double* jac = (which == 0 ? p.m_jac : p.f_jac) + R * n;
  const double* xn = p.x + ((int64_t)b * (T + 1) + t) * nx;
  __shared__ double dx[DDP_MAXJ * 2];
  __shared__ double Jl[36];
  const DevModel& mdl = *p.model;
  if (mdl.ff) {
    // x_new (-) origin on the group, and d(x_new (-) origin)/dx_new = blockdiag(Jlog6, I) (mat_seq_common.hpp:80-86,
    // problem.hpp:414-439 with model_t::d_difference_dq_finish, pinocchio_model.ipp:306-321)
    if (threadIdx.x == 0) {
      lie::difference_x(mdl, org, xn, dx);
      lie::se3_Jlog(dx, Jl);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += OBS) dx[i] = xn[i] - org[i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < r; i += OBS) {
    double s = 0.0;
    for (int l = 0; l < n; ++l) s += jac[i + (int64_t)l * r] * dx[l];
    val[i] += s;
  }
  if (mdl.ff) {
    // jac <- jac * blockdiag(Jlog6, I): only the six leading columns change; one lane per row
    double* jacw = const_cast<double*>(jac);
    for (int i = threadIdx.x; i < r; i += OBS) {
      double row[6], out[6];
      for (int l = 0; l < 6; ++l) row[l] = jacw[i + (int64_t)l * r];
      for (int c = 0; c < 6; ++c) {
        double s = 0.0;
        for (int l = 0; l < 6; ++l) s += row[l] * Jl[6 * l + c];
        out[c] = s;
      }
      for (int c = 0; c < 6; ++c) jacw[i + (int64_t)c * r] = out[c];
    }
  }
  for (int i = threadIdx.x; i < nx; i += OBS) org[i] = xn[i];
}

// optimality_constr = max_t ||eq_t||, optimality_obj = max_t ||l_u + eq_u^T (pe + mu eq) + f_u^T adj|| with the
// adjoint adj_t = f_x^T adj_{t+1} + l_x + eq_x^T (mu eq + pe) + pe_x^T eq, adj_T = l_fx (ddp.hpp:576-627)
__global__ __launch_bounds__(OBS) void optimality_kernel(OuterParams p) {
  const int b = blockIdx.x;
  const int64_t T = p.d.T;
  const int n = (int)p.d.n, nx = (int)p.d.nx, m = (int)p.d.m;
  const int tid = threadIdx.x;
  const double mu = p.mu[b];
  __shared__ double adj[DDP_MAXJ * 2], adj2[DDP_MAXJ * 2], dxv[DDP_MAXJ * 2], pe[DDP_MAXJ * 2], lu[DDP_MAXJ], eqs[DDP_MAXJ * 2];
  for (int i = tid; i < n; i += OBS) adj[i] = p.lfx[(int64_t)b * n + i];
  double obj = 0.0, constr = 0.0;
  __syncthreads();
  for (int64_t t = T - 1; t >= 0; --t) {
    const int e = (int)p.ne[t];
    const int64_t E = (int64_t)b * p.d.Etot + p.Epre[t];
    const int64_t bt = (int64_t)b * T + t;
    const double* eqv = p.eq_val + E;
    const double* eqx = p.eq_x + E * n;
    const double* equ = p.eq_u + E * m;
    const double* jac = p.m_jac + E * n;
    const double* fx = p.fx + bt * n * n;
    const double* fu = p.fu + bt * n * m;
    const double* xt = p.x + ((int64_t)b * (T + 1) + t) * nx;
    const double* org = p.m_origin + bt * nx;
    if (p.model->ff) { if (tid == 0) lie::difference_x(*p.model, org, xt, dxv); }
    else for (int i = tid; i < n; i += OBS) dxv[i] = xt[i] - org[i];
    for (int i = tid; i < e; i += OBS) eqs[i] = eqv[i];
    __syncthreads();
    for (int i = tid; i < e; i += OBS) {          // pe = val + jac (x - origin)
      double s = 0.0;
      for (int l = 0; l < n; ++l) s += jac[i + (int64_t)l * e] * dxv[l];
      pe[i] = p.m_val[E + i] + s;
    }
    __syncthreads();
    for (int j = tid; j < m; j += OBS) {
      double v = p.lu[bt * m + j];
      double s = 0.0;
      for (int i = 0; i < e; ++i) s += equ[i + (int64_t)j * e] * pe[i];
      v += s;
      s = 0.0;
      for (int i = 0; i < e; ++i) s += mu * eqs[i] * equ[i + (int64_t)j * e];
      v += s;
      s = 0.0;
      for (int l = 0; l < n; ++l) s += fu[l + (int64_t)j * n] * adj[l];
      v += s;
      lu[j] = v;
    }
    __syncthreads();
    if (tid == 0) {                               // norms in index order, like the reference's .norm() on a short vector
      double nr = 0.0;
      for (int j = 0; j < m; ++j) nr += lu[j] * lu[j];
      nr = sqrt(nr);
      if (nr > obj) obj = nr;
      double s = 0.0;
      for (int i = 0; i < e; ++i) s += eqs[i] * eqs[i];
      s = sqrt(s);
      if (s > constr) constr = s;
    }
    for (int j = tid; j < n; j += OBS) {
      double s = 0.0;
      for (int l = 0; l < n; ++l) s += fx[l + (int64_t)j * n] * adj[l];
      double v = 0.0 + s;
      v += p.lx[bt * n + j];
      s = 0.0;
      for (int i = 0; i < e; ++i) s += mu * eqs[i] * eqx[i + (int64_t)j * e];
      v += s;
      s = 0.0;
      for (int i = 0; i < e; ++i) s += eqx[i + (int64_t)j * e] * pe[i];
      v += s;
      s = 0.0;
      for (int i = 0; i < e; ++i) s += jac[i + (int64_t)j * e] * eqs[i];
      v += s;
      adj2[j] = v;
    }
    __syncthreads();
    for (int j = tid; j < n; j += OBS) adj[j] = adj2[j];
    __syncthreads();
  }
  if (tid == 0) { p.out[2 * b] = obj; p.out[2 * b + 1] = constr; }
}

// ddp.hpp:680-688
__global__ __launch_bounds__(OBS) void update_multipliers_kernel(OuterParams p) {
  const int64_t t = blockIdx.x, T = p.d.T;
  const int b = blockIdx.y;
  const int n = (int)p.d.n, m = (int)p.d.m;
  const int e = (int)p.ne[t];
  if (e == 0) return;
  const double mu = p.mu[b];
  const int64_t E = (int64_t)b * p.d.Etot + p.Epre[t];
  const int64_t bt = (int64_t)b * T + t;
  const double* eqv = p.eq_val + E;
  const double* eqx = p.eq_x + E * n;
  const double* equ = p.eq_u + E * m;
  const double* k = p.f_val + bt * m;
  const double* K = p.f_jac + bt * m * n;
  for (int i = threadIdx.x; i < e; i += OBS) {
    double s = eqv[i];
    for (int l = 0; l < m; ++l) s += equ[i + (int64_t)l * e] * k[l];
    p.m_val[E + i] += mu * s;
  }
  for (int idx = threadIdx.x; idx < e * n; idx += OBS) {
    const int i = idx % e, j = idx / e;
    double s = eqx[i + (int64_t)j * e];
    for (int l = 0; l < m; ++l) s += equ[i + (int64_t)l * e] * K[l + (int64_t)j * m];
    p.m_jac[E * n + idx] += mu * s;
  }
}

OuterParams make_params(ddp_hip_ctx* ctx) {
  OuterParams p{};
  p.d = ctx->d;
  p.model = ctx->model_d;
  p.ne = ctx->ne_d;
  p.Epre = ctx->Epre_d;
  auto S = [&](int s) { return ctx->seq[s].ptr; };
  p.x = S(DDP_HIP_SEQ_X);
  p.lfx = S(DDP_HIP_SEQ_LFX); p.lx = S(DDP_HIP_SEQ_LX); p.lu = S(DDP_HIP_SEQ_LU);
  p.fx = S(DDP_HIP_SEQ_FX); p.fu = S(DDP_HIP_SEQ_FU);
  p.eq_val = S(DDP_HIP_SEQ_EQ_VAL); p.eq_x = S(DDP_HIP_SEQ_EQ_X); p.eq_u = S(DDP_HIP_SEQ_EQ_U);
  p.m_origin = S(DDP_HIP_SEQ_MULT_ORIGIN); p.m_val = S(DDP_HIP_SEQ_MULT_VAL); p.m_jac = S(DDP_HIP_SEQ_MULT_JAC);
  p.f_origin = S(DDP_HIP_SEQ_FB_ORIGIN); p.f_val = S(DDP_HIP_SEQ_FB_VAL); p.f_jac = S(DDP_HIP_SEQ_FB_JAC);
  p.mu = ctx->mu_d;
  p.out = ctx->ws_Q;      // two doubles per instance of the backward workspace, idle between sweeps
  return p;
}

}  // namespace

extern "C" int ddp_hip_update_origin(ddp_hip_ctx* ctx, int which) {
  if (!ctx || (which != 0 && which != 1)) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  if (which == 0 && ctx->d.Etot == 0) {
    // no rows: only the origins move
  }
  OuterParams p = make_params(ctx);
  hipLaunchKernelGGL(update_origin_kernel, dim3((unsigned)ctx->d.T, (unsigned)ctx->d.batch), dim3(OBS), 0, ctx->stream, p, which);
  HIP_TRY(hipGetLastError());
  END_SYNC(ctx);
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_optimality(ddp_hip_ctx* ctx, const double* mu, double* obj_out, double* constr_out) {
  if (!ctx || !mu || !obj_out || !constr_out) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  const int64_t B = ctx->d.batch;
  HIP_TRY(hipMemcpyAsync(ctx->mu_d, mu, sizeof(double) * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
  OuterParams p = make_params(ctx);
  hipLaunchKernelGGL(optimality_kernel, dim3((unsigned)B), dim3(OBS), 0, ctx->stream, p);
  HIP_TRY(hipGetLastError());
  std::vector<double> out((size_t)(2 * B));
  HIP_TRY(hipMemcpyAsync(out.data(), p.out, sizeof(double) * (size_t)(2 * B), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int64_t b = 0; b < B; ++b) { obj_out[b] = out[(size_t)(2 * b)]; constr_out[b] = out[(size_t)(2 * b + 1)]; }
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_update_multipliers(ddp_hip_ctx* ctx, const double* mu) {
  if (!ctx || !mu) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->d.Etot == 0) return DDP_HIP_OK;
  HIP_TRY(hipMemcpyAsync(ctx->mu_d, mu, sizeof(double) * (size_t)ctx->d.batch, hipMemcpyHostToDevice, ctx->stream));
  OuterParams p = make_params(ctx);
  hipLaunchKernelGGL(update_multipliers_kernel, dim3((unsigned)ctx->d.T, (unsigned)ctx->d.batch), dim3(OBS), 0, ctx->stream, p);
  HIP_TRY(hipGetLastError());
  END_SYNC(ctx);
  return DDP_HIP_OK;
}
