// lin_common.h -- parameter block shared by the linearisation translation units (lin.hip, lin_static.hip)
#pragma once
#include "internal.h"

struct LinParams {
  Dims d;
  const DevModel* model;
  const int64_t* ne;
  const int64_t* Epre;
  const double* target;
  const double *x, *u;
  double *lfx, *lfxx, *lx, *lu, *lxx, *lux, *luu;
  double *f_val, *fx, *fu, *fxx, *fux, *fuu;
  double *eq_val, *eq_x, *eq_u, *eq_xx, *eq_ux, *eq_uu;
  int32_t has_tensors;
  double *eq_xk, *eq_fxk, *eq_c;   // large-model constraint chain workspace: x_1..x_K | f_x(x_1..x_{K-1}) | base jacobian
  double* vcache;   // [batch*T][2nv+1][nv*VC_STRIDE]: (q, v)-dependent part at (q,v), (q, v+eps e_i), (q+eps e_i, v)
  double* qcache;   // [batch*T][nv+1][nv*QC_STRIDE]: q-dependent part of the ABA at the base q and at q + eps e_i (mode 2)
  int32_t skip_qv_mirror;   // static stencil: do not write the mirror images f_xx(:, i, j), f_uu(:, i, j), i < j, of the symmetric pairs --
                            // the backward sweep never reads them when it knows the tensors symmetric (bwd_split.h: only the columns
                            // j >= c of slab c); formed on demand for any other reader (lin.hip: lin_materialize_fxx)
  int32_t skip_top;         // static stencil: the configuration rows k < nv of a tensor column are exact zeros except at k = (first
                            // direction) mod nv and k = (second direction) mod nv -- q+ = q + dt v is affine, and row k of every stencil
                            // point but those two is bit for bit the base point's -- so only those two and the rows k >= nv are formed
                            // and stored; the other rows hold the zeros lin.hip: tensor_zero_top_kernel left there
  double* accel_out;     // static first-order kernels, levels 1 and 2: when set, the wave stores the accelerations of its nv perturbed
                         // points, [pair][3 nv directions: q, v, u | the unperturbed point][nv], instead of the jacobian columns (lin_analytic.hip: mode 1 reads them)
  int32_t ncfg, nvcfg;   // entries per (instance, t) of the q- / v-cache: nv+1 / 2nv+1 with the mode-2 stencil resident,
                         // 1 / 1 when only the first order is formed (tensor-free contexts: base configuration and base (q, v))
};

constexpr int LBS = 64;

// static-topology second-order path (lin_static.hip): returns non-zero when the model's tree matches a compiled-in
// topology; the launcher covers the velocity- and torque-level stencil points of finite_diff_hessian_compute mode 2
int lin_static_supported(const DevModel& m);   // 0: none, else the id of the compiled-in topology
int lin_static_launch(ddp_hip_ctx* ctx, const LinParams& p, int level);   // DDP_HIP_OK or an error code
int64_t lin_static_ws_per_bt(const DevModel& m);

// analytic first order / mode-1 second order of large tree models (lin_analytic.hip): stage 0 = f_x, f_u at the
// trajectory points (problem.hpp:463-503), stage 1 = forward differences of those jacobians (problem.hpp:67-150)
int lin_analytic_setup(ddp_hip_ctx* ctx);
void lin_analytic_teardown(ddp_hip_ctx* ctx);
// flags: LIN_ANA_F the dynamics' outputs, LIN_ANA_EQ the constraint chain's (problem.hpp:569-620 on the analytic jacobians)
constexpr int LIN_ANA_F = 1, LIN_ANA_EQ = 2;
constexpr int LIN_ANA_ACCEL = 4;   // stage 0: the mode-1 pass follows in this linearisation call -- form its accelerations now and take the trajectory point's from them
int lin_analytic_launch(ddp_hip_ctx* ctx, const LinParams& p, int stage, int flags);
