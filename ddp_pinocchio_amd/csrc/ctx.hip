// ctx.hip -- context life cycle, resident sequences, profiling.  gfx950 only.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <limits>
#include <new>

#include "internal.h"

extern "C" int ddp_hip_abi_version(void) { return DDP_HIP_ABI_VERSION; }

extern "C" const char* ddp_hip_strerror(int code) {
  switch (code) {
    case DDP_HIP_OK: return "ok";
    case DDP_HIP_EV_LLT_RESTART: return "backward sweep restarted after a non-positive LLT pivot";
    case DDP_HIP_EV_LINESEARCH_FLOOR: return "line search reached step < 1e-10";
    case DDP_HIP_E_ARG: return "invalid argument";
    case DDP_HIP_E_HIP: return "HIP runtime error";
    case DDP_HIP_E_NODEVICE: return "no HIP device (there is no CPU fallback)";
    case DDP_HIP_E_UNSUPPORTED: return "unsupported configuration";
    case DDP_HIP_E_MAX_RESTARTS: return "backward sweep exceeded max_restarts";
    case DDP_HIP_E_COMM: return "RCCL error";
    default: return "unknown";
  }
}

extern "C" int ddp_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

static void pack_body_inertia(const ddp_hip_model* m, int i, double* out21) {
  // spatial inertia [Ic + m cx cx^T, m cx; m cx^T, m 1] (Featherstone RBDA eq. 2.63), lower triangle
  double I6[6][6];
  memset(I6, 0, sizeof(I6));
  const double* c = m->com + 3 * i;
  double mass = m->mass_j[i];
  double cx[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < 3; ++l) {
      double cc = 0;
      for (int j = 0; j < 3; ++j) cc += cx[k][j] * cx[l][j];
      I6[k][l] = m->Ic[9 * i + 3 * k + l] + mass * cc;
      I6[k][l + 3] = mass * cx[k][l];
      I6[k + 3][l] = mass * cx[l][k];
    }
  I6[3][3] = I6[4][4] = I6[5][5] = mass;
  int p = 0;
  for (int r = 0; r < 6; ++r)
    for (int cc2 = 0; cc2 <= r; ++cc2) out21[p++] = I6[r][cc2];
}

// Slot tables for the small-state tree traversals (rbd.h: aba_u_cached).  Leaf->root (descending index): a joint's
// accumulator is allocated when its largest-index child contributes and freed once the joint itself is processed.
// Root->leaf (ascending index): a joint's value is kept from when it is processed until its largest-index child
// has read it.  For a humanoid tree a handful of slots suffice.
static bool build_slot_tables(DevModel& dm) {
  const int N = dm.nj;
  int largest_child[DDP_MAXJ];
  for (int i = 0; i < N; ++i) { dm.has_child[i] = 0; largest_child[i] = -1; dm.slot_up[i] = -1; dm.slot_down[i] = -1; }
  for (int i = 0; i < N; ++i)
    if (dm.parent[i] >= 0) { dm.has_child[dm.parent[i]] = 1; if (i > largest_child[dm.parent[i]]) largest_child[dm.parent[i]] = i; }
  for (int i = 0; i < N; ++i) {
    const int par = dm.parent[i];
    dm.first_contrib[i] = dm.last_child[i] = (par >= 0 && largest_child[par] == i) ? 1 : 0;
  }
  const int MAXS = 16;
  bool used[MAXS];
  int n_slots = 0;
  for (int k = 0; k < MAXS; ++k) used[k] = false;
  for (int i = N - 1; i >= 0; --i) {            // leaf -> root
    const int par = dm.parent[i];
    if (par >= 0 && dm.first_contrib[i]) {
      int k = 0;
      while (k < MAXS && used[k]) ++k;
      if (k == MAXS) return false;
      used[k] = true;
      dm.slot_up[par] = k;
      if (k + 1 > n_slots) n_slots = k + 1;
    }
    if (dm.slot_up[i] >= 0) used[dm.slot_up[i]] = false;   // the joint is processed: its accumulator is consumed
  }
  for (int k = 0; k < MAXS; ++k) used[k] = false;
  for (int i = 0; i < N; ++i) {                 // root -> leaf
    if (dm.has_child[i]) {
      int k = 0;
      while (k < MAXS && used[k]) ++k;
      if (k == MAXS) return false;
      used[k] = true;
      dm.slot_down[i] = k;
      if (k + 1 > n_slots) n_slots = k + 1;
    }
    const int par = dm.parent[i];
    if (par >= 0 && dm.last_child[i]) used[dm.slot_down[par]] = false;
  }
  dm.n_slots = n_slots;
  // level schedule + children lists
  int level[DDP_MAXJ];
  int nl = 0;
  for (int i = 0; i < N; ++i) { level[i] = dm.parent[i] >= 0 ? level[dm.parent[i]] + 1 : 0; if (level[i] + 1 > nl) nl = level[i] + 1; }
  dm.n_levels = nl;
  dm.max_level_width = 0;
  int pos = 0;
  for (int L = 0; L < nl; ++L) {
    dm.lvl_start[L] = pos;
    for (int i = 0; i < N; ++i) if (level[i] == L) dm.lvl_joint[pos++] = i;
    if (pos - dm.lvl_start[L] > dm.max_level_width) dm.max_level_width = pos - dm.lvl_start[L];
  }
  dm.lvl_start[nl] = pos;
  pos = 0;
  for (int i = 0; i < N; ++i) {
    dm.child_start[i] = pos;
    for (int c = N - 1; c > i; --c) if (dm.parent[c] == i) dm.child_list[pos++] = c;
  }
  dm.child_start[N] = pos;
  return n_slots <= 8;   // rbd::MAX_SLOTS
}

// the model table of the C-ABI as the device-side DevModel (shared with model_api.hip)
void ddp_hip_fill_dev_model(const ddp_hip_model* mo, DevModel& dm) {
  memset(&dm, 0, sizeof(dm));
  dm.kind = mo->kind; dm.nv = mo->nv; dm.mass = mo->mass; dm.length = mo->length;
  dm.ff = (mo->kind == DDP_HIP_MODEL_TREE && mo->jtype && mo->jtype[0] == DDP_HIP_JOINT_FREEFLYER) ? 1 : 0;
  dm.nj = dm.ff ? mo->nv - 5 : mo->nv;
  dm.nq = dm.ff ? mo->nv + 1 : mo->nv;
  for (int k = 0; k < 3; ++k) dm.gravity[k] = mo->gravity[k];
  if (mo->kind != DDP_HIP_MODEL_TREE) return;
  for (int i = 0; i < dm.nj; ++i) {
    dm.parent[i] = mo->parent[i]; dm.jtype[i] = mo->jtype[i];
    for (int k = 0; k < 3; ++k) { dm.axis[i][k] = mo->axis[3 * i + k]; dm.pp[i][k] = mo->pp[3 * i + k]; }
    for (int k = 0; k < 9; ++k) dm.Rp[i][k] = mo->Rp[9 * i + k];
    pack_body_inertia(mo, i, dm.I6[i]);
  }
}
bool ddp_hip_build_tables(DevModel& dm) { return build_slot_tables(dm); }

static int64_t seq_size_of(const Dims& d, int s) {
  const int64_t T = d.T, n = d.n, m = d.m, nx = d.nx, E = d.Etot;
  switch (s) {
    case DDP_HIP_SEQ_X: case DDP_HIP_SEQ_X_NEW: return (T + 1) * nx;
    case DDP_HIP_SEQ_U: case DDP_HIP_SEQ_U_NEW: return T * m;
    case DDP_HIP_SEQ_LFX: return n;
    case DDP_HIP_SEQ_LFXX: return n * n;
    case DDP_HIP_SEQ_LX: return T * n;
    case DDP_HIP_SEQ_LU: return T * m;
    case DDP_HIP_SEQ_LXX: return T * n * n;
    case DDP_HIP_SEQ_LUX: return T * m * n;
    case DDP_HIP_SEQ_LUU: return T * m * m;
    case DDP_HIP_SEQ_F_VAL: return T * nx;
    case DDP_HIP_SEQ_FX: return T * n * n;
    case DDP_HIP_SEQ_FU: return T * n * m;
    case DDP_HIP_SEQ_FXX: return T * n * n * n;
    case DDP_HIP_SEQ_FUX: return T * n * m * n;
    case DDP_HIP_SEQ_FUU: return T * n * m * m;
    case DDP_HIP_SEQ_EQ_VAL: return E;
    case DDP_HIP_SEQ_EQ_X: return E * n;
    case DDP_HIP_SEQ_EQ_U: return E * m;
    case DDP_HIP_SEQ_EQ_XX: return E * n * n;
    case DDP_HIP_SEQ_EQ_UX: return E * m * n;
    case DDP_HIP_SEQ_EQ_UU: return E * m * m;
    case DDP_HIP_SEQ_MULT_ORIGIN: return T * nx;
    case DDP_HIP_SEQ_MULT_VAL: return E;
    case DDP_HIP_SEQ_MULT_JAC: return E * n;
    case DDP_HIP_SEQ_FB_ORIGIN: return T * nx;
    case DDP_HIP_SEQ_FB_VAL: return T * m;
    case DDP_HIP_SEQ_FB_JAC: return T * m * n;
    case DDP_HIP_SEQ_VX_TRACE: return T * n;
    case DDP_HIP_SEQ_VXX_TRACE: return T * n * n;
    case DDP_HIP_SEQ_COSTS_OLD: case DDP_HIP_SEQ_COSTS_NEW: return T + 1;
    default: return -1;
  }
}

static bool is_tensor_seq(int s) {
  return s == DDP_HIP_SEQ_FXX || s == DDP_HIP_SEQ_FUX || s == DDP_HIP_SEQ_FUU || s == DDP_HIP_SEQ_EQ_XX ||
         s == DDP_HIP_SEQ_EQ_UX || s == DDP_HIP_SEQ_EQ_UU;
}
static bool is_trace_seq(int s) { return s == DDP_HIP_SEQ_VX_TRACE || s == DDP_HIP_SEQ_VXX_TRACE; }
static bool nan_init_seq(int s) {
  // mat_seq_t storage is NaN-poisoned at construction (detail/mat_seq.hpp:34-37); uninit_derivative_storage
  // then zeroes every derivative sequence except f_val, and leaves lfx / lfxx NaN (ddp.hpp:441-442,476-511)
  return s == DDP_HIP_SEQ_LFX || s == DDP_HIP_SEQ_LFXX || s == DDP_HIP_SEQ_F_VAL || s == DDP_HIP_SEQ_FB_ORIGIN ||
         s == DDP_HIP_SEQ_FB_VAL || s == DDP_HIP_SEQ_FB_JAC || s == DDP_HIP_SEQ_X || s == DDP_HIP_SEQ_U ||
         s == DDP_HIP_SEQ_X_NEW || s == DDP_HIP_SEQ_U_NEW || is_trace_seq(s);
}

__global__ void fill_kernel(double* p, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

static int fill_device(ddp_hip_ctx* ctx, double* p, int64_t n, double v) {
  if (n <= 0) return DDP_HIP_OK;
  if (v == 0.0) {
    HIP_TRY(hipMemsetAsync(p, 0, (size_t)n * sizeof(double), ctx->stream));
    return DDP_HIP_OK;
  }
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, n, v);
  HIP_TRY(hipGetLastError());
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_create(const ddp_hip_problem* prob, int device, uint32_t flags, ddp_hip_ctx** out) {
  if (!prob || !out) return DDP_HIP_E_ARG;
  *out = nullptr;
  const ddp_hip_model& mo = prob->model;
  if (prob->T < 1 || prob->batch < 1 || mo.nv < 1 || mo.nv > DDP_MAXJ) return DDP_HIP_E_ARG;
  if (mo.kind != DDP_HIP_MODEL_PENDULUM && mo.kind != DDP_HIP_MODEL_TREE) return DDP_HIP_E_ARG;
  if (mo.kind == DDP_HIP_MODEL_PENDULUM && mo.nv != 1) return DDP_HIP_E_ARG;
  if (prob->fd_mode < 0 || prob->fd_mode > 2) return DDP_HIP_E_ARG;
  if (prob->eq_kind != DDP_HIP_EQ_NONE && (!prob->ne || prob->eq_advance < 0 || prob->eq_advance > 4)) return DDP_HIP_E_ARG;
  if (prob->eq_kind == DDP_HIP_EQ_FRAME && (mo.kind != DDP_HIP_MODEL_TREE || prob->frame_joint < 0 ||
                                            prob->frame_joint >= (mo.jtype && mo.jtype[0] == DDP_HIP_JOINT_FREEFLYER ? mo.nv - 5 : mo.nv)))
    return DDP_HIP_E_ARG;
  bool ff = false;
  if (mo.kind == DDP_HIP_MODEL_TREE) {
    if (!mo.parent || !mo.jtype || !mo.axis || !mo.Rp || !mo.pp || !mo.mass_j || !mo.com || !mo.Ic) return DDP_HIP_E_ARG;
    ff = mo.jtype[0] == DDP_HIP_JOINT_FREEFLYER;
    const int nj = ff ? mo.nv - 5 : mo.nv;
    if (nj < 1 || (ff && mo.parent[0] != -1)) return DDP_HIP_E_ARG;
    for (int i = 0; i < nj; ++i) {
      if (mo.parent[i] >= i || mo.parent[i] < -1) return DDP_HIP_E_ARG;
      if (i > 0 && mo.jtype[i] != DDP_HIP_JOINT_REVOLUTE && mo.jtype[i] != DDP_HIP_JOINT_PRISMATIC) return DDP_HIP_E_ARG;   // one free flyer, at the root
    }
    // Lie-group configurations: forward-differenced jacobians (the north star) and mode 2 / tensor-free only -- the
    // reference's mode 1 asserts nq == nv itself (problem.hpp:78-81); the config constraint subtracts configurations
    // (problem.hpp:785-790), meaningless on a quaternion: frame constraints only
    if (ff && (!prob->first_order_fd || prob->fd_mode == 1 || prob->eq_kind == DDP_HIP_EQ_CONFIG)) return DDP_HIP_E_UNSUPPORTED;
  }
  int ndev = ddp_hip_device_count();
  if (ndev <= 0) return DDP_HIP_E_NODEVICE;
  if (device < 0 || device >= ndev) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(device));

  ddp_hip_ctx* ctx = new (std::nothrow) ddp_hip_ctx();
  if (!ctx) return DDP_HIP_E_HIP;
  ctx->device = device;
  ctx->flags = flags;
  ctx->active_h.assign((size_t)prob->batch, 1);
  Dims& d = ctx->d;
  d.T = prob->T; d.nv = mo.nv; d.n = 2 * (int64_t)mo.nv; d.m = mo.nv; d.nx = 2 * (int64_t)mo.nv + (ff ? 1 : 0); d.batch = prob->batch;
  ctx->ne_h.assign((size_t)d.T, 0);
  ctx->Epre_h.assign((size_t)d.T + 1, 0);
  d.emax = 0;
  for (int64_t t = 0; t < d.T; ++t) {
    int64_t e = (prob->eq_kind != DDP_HIP_EQ_NONE && prob->ne) ? prob->ne[t] : 0;
    if (e < 0) { delete ctx; return DDP_HIP_E_ARG; }
    if (prob->eq_kind == DDP_HIP_EQ_CONFIG && e != 0 && e != mo.nv) { delete ctx; return DDP_HIP_E_ARG; }
    if (prob->eq_kind == DDP_HIP_EQ_FRAME && e != 0 && e != 3) { delete ctx; return DDP_HIP_E_ARG; }
    ctx->ne_h[(size_t)t] = e;
    ctx->Epre_h[(size_t)t + 1] = ctx->Epre_h[(size_t)t] + e;
    if (e > d.emax) d.emax = e;
  }
  d.Etot = ctx->Epre_h[(size_t)d.T];

#define CTX_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) { (void)hipGetLastError(); ddp_hip_destroy(ctx); return DDP_HIP_E_HIP; } \
  } while (0)

  CTX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  CTX_TRY(hipMalloc(&ctx->ne_d, sizeof(int64_t) * (size_t)d.T));
  CTX_TRY(hipMalloc(&ctx->Epre_d, sizeof(int64_t) * (size_t)(d.T + 1)));
  CTX_TRY(hipMemcpy(ctx->ne_d, ctx->ne_h.data(), sizeof(int64_t) * (size_t)d.T, hipMemcpyHostToDevice));
  CTX_TRY(hipMemcpy(ctx->Epre_d, ctx->Epre_h.data(), sizeof(int64_t) * (size_t)(d.T + 1), hipMemcpyHostToDevice));
  if (d.Etot > 0) {
    if (!prob->eq_target) { ddp_hip_destroy(ctx); return DDP_HIP_E_ARG; }
    CTX_TRY(hipMalloc(&ctx->target_d, sizeof(double) * (size_t)d.Etot));
    CTX_TRY(hipMemcpy(ctx->target_d, prob->eq_target, sizeof(double) * (size_t)d.Etot, hipMemcpyHostToDevice));
  }

  DevModel& dm = ctx->model_h;
  ddp_hip_fill_dev_model(&mo, dm);
  for (int k = 0; k < 3; ++k) dm.frame_off[k] = prob->frame_off[k];
  dm.dt = prob->dt; dm.c = prob->c;
  dm.eq_kind = prob->eq_kind; dm.eq_advance = prob->eq_advance; dm.frame_joint = prob->frame_joint;
  dm.first_order_fd = prob->first_order_fd; dm.fd_mode = prob->fd_mode;
  if (mo.kind == DDP_HIP_MODEL_TREE && !build_slot_tables(dm)) { ddp_hip_destroy(ctx); return DDP_HIP_E_UNSUPPORTED; }
  CTX_TRY(hipMalloc(&ctx->model_d, sizeof(DevModel)));
  CTX_TRY(hipMemcpy(ctx->model_d, &dm, sizeof(DevModel), hipMemcpyHostToDevice));

  const double qnan = std::numeric_limits<double>::quiet_NaN();
  for (int s = 0; s < DDP_HIP_SEQ_COUNT; ++s) {
    int64_t sz = seq_size_of(d, s);
    ctx->seq[s].size = sz;
    if ((flags & DDP_HIP_FLAG_NO_TENSORS) && is_tensor_seq(s)) continue;
    if (!(flags & DDP_HIP_FLAG_TRACE) && is_trace_seq(s)) continue;
    if (sz <= 0) continue;
    CTX_TRY(hipMalloc(&ctx->seq[s].ptr, sizeof(double) * (size_t)(sz * d.batch)));
    if (fill_device(ctx, ctx->seq[s].ptr, sz * d.batch, nan_init_seq(s) ? qnan : 0.0) != DDP_HIP_OK) {
      ddp_hip_destroy(ctx);
      return DDP_HIP_E_HIP;
    }
  }
  int rc = bwd_setup(ctx);
  if (rc == DDP_HIP_OK) rc = fwd_setup(ctx);
  if (rc == DDP_HIP_OK) rc = lin_setup(ctx);
  if (rc != DDP_HIP_OK) { ddp_hip_destroy(ctx); return rc; }
  CTX_TRY(hipStreamSynchronize(ctx->stream));
#undef CTX_TRY
  *out = ctx;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_destroy(ddp_hip_ctx* ctx) {
  if (!ctx) return DDP_HIP_E_ARG;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  lin_teardown(ctx);
  fwd_teardown(ctx);
  bwd_teardown(ctx);
  for (int s = 0; s < DDP_HIP_SEQ_COUNT; ++s)
    if (ctx->seq[s].ptr) (void)hipFree(ctx->seq[s].ptr);
  if (ctx->ne_d) (void)hipFree(ctx->ne_d);
  if (ctx->Epre_d) (void)hipFree(ctx->Epre_d);
  if (ctx->target_d) (void)hipFree(ctx->target_d);
  if (ctx->model_d) (void)hipFree(ctx->model_d);
  for (int k = 0; k < DDP_HIP_K_COUNT; ++k) {
    for (auto e : ctx->prof[k].starts) (void)hipEventDestroy(e);
    for (auto e : ctx->prof[k].stops) (void)hipEventDestroy(e);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return DDP_HIP_OK;
}

extern "C" void* ddp_hip_stream(ddp_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int ddp_hip_synchronize(ddp_hip_ctx* ctx) {
  if (!ctx) return DDP_HIP_E_ARG;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DDP_HIP_OK;
}

extern "C" int64_t ddp_hip_batch(const ddp_hip_ctx* ctx) { return ctx ? ctx->d.batch : -1; }

extern "C" int64_t ddp_hip_seq_size(const ddp_hip_ctx* ctx, int seq) {
  if (!ctx || seq < 0 || seq >= DDP_HIP_SEQ_COUNT) return -1;
  return ctx->seq[seq].size;
}

extern "C" double* ddp_hip_device_ptr(ddp_hip_ctx* ctx, int seq) {
  if (!ctx || seq < 0 || seq >= DDP_HIP_SEQ_COUNT) return nullptr;
  if (seq == DDP_HIP_SEQ_FXX || seq == DDP_HIP_SEQ_FUX || seq == DDP_HIP_SEQ_FUU) { ctx->tensor_tops_zero = false; ctx->tensor_tops_sparse = false; ctx->fuu_zero = false; }
  if (seq == DDP_HIP_SEQ_FXX || seq == DDP_HIP_SEQ_FUU) {
    (void)hipSetDevice(ctx->device);
    if (lin_materialize_fxx(ctx) != DDP_HIP_OK) return nullptr;
    ctx->tensors_sym = false;                              // the caller may write through the pointer: K3 reads every half-slab again
  }
  return ctx->seq[seq].ptr;
}

static int check_range(ddp_hip_ctx* ctx, int seq, int64_t first, int64_t count) {
  if (!ctx || seq < 0 || seq >= DDP_HIP_SEQ_COUNT) return DDP_HIP_E_ARG;
  if (first < 0 || count < 0 || first + count > ctx->d.batch) return DDP_HIP_E_ARG;
  if (ctx->seq[seq].size > 0 && !ctx->seq[seq].ptr) return DDP_HIP_E_UNSUPPORTED;  // not allocated under the create flags
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_upload(ddp_hip_ctx* ctx, int seq, const double* host, int64_t first, int64_t count) {
  int rc = check_range(ctx, seq, first, count);
  if (rc != DDP_HIP_OK) return rc;
  int64_t sz = ctx->seq[seq].size;
  if (sz == 0 || count == 0) return DDP_HIP_OK;
  if (!host) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  if (seq == DDP_HIP_SEQ_FXX || seq == DDP_HIP_SEQ_FUX || seq == DDP_HIP_SEQ_FUU) { ctx->tensor_tops_zero = false; ctx->tensor_tops_sparse = false; ctx->fuu_zero = false; }
  if (seq == DDP_HIP_SEQ_FXX || seq == DDP_HIP_SEQ_FUU) {   // tensors from outside: no symmetry assumed (bwd_split.h)
    { const int rc_ = lin_materialize_fxx(ctx); if (rc_ != DDP_HIP_OK) return rc_; }   // what is not overwritten (other instances, the other tensor) stays whole
    ctx->tensors_sym = false;
  }
  HIP_TRY(hipMemcpyAsync(ctx->seq[seq].ptr + first * sz, host, sizeof(double) * (size_t)(sz * count), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_download(ddp_hip_ctx* ctx, int seq, double* host, int64_t first, int64_t count) {
  int rc = check_range(ctx, seq, first, count);
  if (rc != DDP_HIP_OK) return rc;
  int64_t sz = ctx->seq[seq].size;
  if (sz == 0 || count == 0) return DDP_HIP_OK;
  if (!host) return DDP_HIP_E_ARG;
  HIP_TRY(hipSetDevice(ctx->device));
  if (seq == DDP_HIP_SEQ_FXX || seq == DDP_HIP_SEQ_FUU) { const int rc_ = lin_materialize_fxx(ctx); if (rc_ != DDP_HIP_OK) return rc_; }
  HIP_TRY(hipMemcpyAsync(host, ctx->seq[seq].ptr + first * sz, sizeof(double) * (size_t)(sz * count), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_fill(ddp_hip_ctx* ctx, int seq, double value) {
  int rc = check_range(ctx, seq, 0, 0);
  if (rc != DDP_HIP_OK) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  if (seq == DDP_HIP_SEQ_FXX || seq == DDP_HIP_SEQ_FUX || seq == DDP_HIP_SEQ_FUU) { ctx->tensor_tops_zero = false; ctx->tensor_tops_sparse = false; ctx->fuu_zero = false; }
  if (seq == DDP_HIP_SEQ_FXX || seq == DDP_HIP_SEQ_FUU) {
    const int rc_ = lin_materialize_fxx(ctx);
    if (rc_ != DDP_HIP_OK) return rc_;
    ctx->tensors_sym = false;
  }
  return fill_device(ctx, ctx->seq[seq].ptr, ctx->seq[seq].size * ctx->d.batch, value);
}

extern "C" int ddp_hip_set_async(ddp_hip_ctx* ctx, int on) {
  if (!ctx) return DDP_HIP_E_ARG;
  if (!on && ctx->async_mode) { HIP_TRY(hipSetDevice(ctx->device)); HIP_TRY(hipStreamSynchronize(ctx->stream)); }
  ctx->async_mode = on != 0;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_set_active(ddp_hip_ctx* ctx, const int32_t* active) {
  if (!ctx) return DDP_HIP_E_ARG;
  ctx->all_active = true;
  for (int64_t b = 0; b < ctx->d.batch; ++b) {
    ctx->active_h[(size_t)b] = (!active || active[b]) ? 1 : 0;
    if (!ctx->active_h[(size_t)b]) ctx->all_active = false;
  }
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_ctx_info(const ddp_hip_ctx* ctx, ddp_hip_info* out) {
  if (!ctx || !out) return DDP_HIP_E_ARG;
  const Dims& d = ctx->d;
  out->device = ctx->device;
  out->lin_path = ctx->model_h.kind == DDP_HIP_MODEL_PENDULUM ? 0 : (ctx->lin_static ? 1 + ctx->lin_static : 1);
  out->first_order = ctx->model_h.kind == DDP_HIP_MODEL_PENDULUM ? 0 : (ctx->model_h.first_order_fd ? 1 : 2);
  out->bwd_path = (d.n == 76 && d.m == 38 && getenv("DDP_HIP_GENERIC_BWD") == nullptr) ? 1 : 0;
  out->fwd_path = fwd_lat_supported(ctx) ? 1 : 0;
  out->has_tensors = (ctx->flags & DDP_HIP_FLAG_NO_TENSORS) ? 0 : 1;
  int64_t bytes = 0;
  for (int s = 0; s < DDP_HIP_SEQ_COUNT; ++s)
    if (ctx->seq[s].ptr) bytes += 8 * ctx->seq[s].size * d.batch;
  out->hbm_bytes = bytes;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_swap_traj(ddp_hip_ctx* ctx) {
  if (!ctx) return DDP_HIP_E_ARG;
  if (!ctx->all_active) {
    // a frozen instance keeps its trajectory: it is cloned into the buffers that become (X, U)
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t sx = ctx->seq[DDP_HIP_SEQ_X].size, su = ctx->seq[DDP_HIP_SEQ_U].size;
    for (int64_t b = 0; b < ctx->d.batch; ++b) {
      if (ctx->active_h[(size_t)b]) continue;
      HIP_TRY(hipMemcpyAsync(ctx->seq[DDP_HIP_SEQ_X_NEW].ptr + b * sx, ctx->seq[DDP_HIP_SEQ_X].ptr + b * sx, sizeof(double) * (size_t)sx, hipMemcpyDeviceToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->seq[DDP_HIP_SEQ_U_NEW].ptr + b * su, ctx->seq[DDP_HIP_SEQ_U].ptr + b * su, sizeof(double) * (size_t)su, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  // swap(traj, new_traj), ddp.hpp:826: the resident buffers trade places, nothing moves in HBM
  std::swap(ctx->seq[DDP_HIP_SEQ_X].ptr, ctx->seq[DDP_HIP_SEQ_X_NEW].ptr);
  std::swap(ctx->seq[DDP_HIP_SEQ_U].ptr, ctx->seq[DDP_HIP_SEQ_U_NEW].ptr);
  return DDP_HIP_OK;
}

// ---- profiling: HIP events on the context's own stream around every launch of a kernel class --
void prof_begin(ddp_hip_ctx* ctx, int kid, hipStream_t stream) {
  if (!(ctx->profile_mask & (2u << kid))) return;
  if (!stream) stream = ctx->stream;
  ProfSlot& p = ctx->prof[kid];
  if (p.used == p.starts.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); return; }
    p.starts.push_back(a);
    p.stops.push_back(b);
  }
  (void)hipEventRecord(p.starts[p.used], stream);
}
void prof_end(ddp_hip_ctx* ctx, int kid, hipStream_t stream) {
  if (!(ctx->profile_mask & (2u << kid))) return;
  if (!stream) stream = ctx->stream;
  ProfSlot& p = ctx->prof[kid];
  if (p.used >= p.stops.size()) return;
  (void)hipEventRecord(p.stops[p.used], stream);
  ++p.used;
}
static void prof_collect(ddp_hip_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (int g = 0; g < ctx->bwd_groups; ++g)
    if (ctx->bwd_stream[g]) (void)hipStreamSynchronize(ctx->bwd_stream[g]);
  for (int k = 0; k < DDP_HIP_K_COUNT; ++k) {
    ProfSlot& p = ctx->prof[k];
    for (size_t i = 0; i < p.used; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.starts[i], p.stops[i]) == hipSuccess) { p.total_ms += ms; ++p.launches; }
      else (void)hipGetLastError();
    }
    p.used = 0;
  }
}

extern "C" int ddp_hip_profile_enable(ddp_hip_ctx* ctx, int on) {
  if (!ctx) return DDP_HIP_E_ARG;
  if (ctx->profile_mask && !on) prof_collect(ctx);
  ctx->profile_mask = on == 1 ? ~0u : (uint32_t)on;      // 1: every class; else bit (1 + kernel_id) selects a class
  return DDP_HIP_OK;
}
extern "C" int ddp_hip_profile_reset(ddp_hip_ctx* ctx) {
  if (!ctx) return DDP_HIP_E_ARG;
  prof_collect(ctx);
  for (int k = 0; k < DDP_HIP_K_COUNT; ++k) { ctx->prof[k].total_ms = 0; ctx->prof[k].launches = 0; }
  return DDP_HIP_OK;
}
extern "C" int ddp_hip_profile_get(ddp_hip_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches) {
  if (!ctx || kernel_id < 0 || kernel_id >= DDP_HIP_K_COUNT) return DDP_HIP_E_ARG;
  prof_collect(ctx);
  if (total_ms) *total_ms = ctx->prof[kernel_id].total_ms;
  if (launches) *launches = ctx->prof[kernel_id].launches;
  return DDP_HIP_OK;
}

// Bytes of f_xx / f_ux / f_uu the contraction kernel (K3) reads per (instance, step) with the tensors in their current state:
// everything (tensors from outside), the columns j >= c of slab c (symmetric: this context's own mode-2 / zero tensors), or the
// lower halves of those columns plus the (at most) two non-zero entries of each upper half (the static stencil's own tensors)
extern "C" int64_t ddp_hip_bwd_stream_bytes(const ddp_hip_ctx* ctx) {
  if (!ctx) return -1;
  if (ctx->flags & DDP_HIP_FLAG_NO_TENSORS) return 0;
  const int64_t n = ctx->d.n, m = ctx->d.m;
  const bool fast = n == 76 && m == 38 && ctx->d.emax <= 52 && getenv("DDP_HIP_GENERIC_BWD") == nullptr;
  const bool sym = fast && ctx->tensors_sym && ctx->jobs_sym_d && getenv("DDP_HIP_K3_NO_SYM") == nullptr;
  const bool half = sym && ctx->tensor_tops_zero && ctx->tensor_tops_sparse && ctx->jobs_half_d && getenv("DDP_HIP_K3_NO_HALF") == nullptr;
  if (half) {
    const int64_t cxx = n * (n + 1) / 2, cux = n * m, cuu = m * (m + 1) / 2;
    return 8 * ((cxx + cux + cuu) * (n - m) + 2 * cxx - n + cux);       // lower halves + two entries per f_xx column (one on its diagonal), one per f_ux column
  }
  if (fast && !half && ctx->fuu_zero && ctx->model_h.fd_mode == 1 && ctx->jobs_half_d && getenv("DDP_HIP_K3_NO_HALF") == nullptr)
    return 8 * (n * n + n * m) * (n - m);                                // analytic mode 1: the lower halves of f_xx and f_ux, nothing of f_uu
  if (sym) return 8 * (n * (n * (n + 1) / 2) + n * n * m + n * (m * (m + 1) / 2));
  return 8 * (n * n * n + n * n * m + n * m * m);
}

extern "C" int64_t ddp_hip_bwd_algorithmic_bytes(const ddp_hip_ctx* ctx) {
  if (!ctx) return -1;
  const Dims& d = ctx->d;
  const int64_t n = d.n, m = d.m, nx = d.nx, T = d.T;
  // B_bwd = 8 T [(n+m+n^2+mn+m^2) + (n^2+nm) + (n^3+n^2 m+n m^2) + (m+mn+nx)] + eq terms
  int64_t per_step = (n + m + n * n + m * n + m * m) + (n * n + n * m) + (m + m * n + nx);
  if (!(ctx->flags & DDP_HIP_FLAG_NO_TENSORS)) per_step += n * n * n + n * n * m + n * m * m;
  int64_t words = T * per_step;
  for (int64_t t = 0; t < T; ++t) {
    int64_t e = ctx->ne_h[(size_t)t];
    words += (e + e * n + e * m) + (e + e * n);
    if (!(ctx->flags & DDP_HIP_FLAG_NO_TENSORS)) words += e * n * n + e * m * n + e * m * m;
  }
  return 8 * words;
}
