// internal.h -- context layout shared by the translation units of libddp_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ddp_hip/ddp_hip.h"

#define DDP_MAXJ DDP_HIP_MAX_JOINTS

// Device-side model table (tree of 1-DoF joints or the closed-form pendulum); lives in HBM, read
// through the scalar / L2 path by every dynamics kernel.
struct DevModel {
  int32_t kind, nv;              // nv: velocity (tangent) dimension
  int32_t ff, nj, nq, pad0_;     // free-flyer root (SE(3), lie.h); number of joints (nv - 5 with a free flyer, else nv); configuration dimension
  double mass, length;
  double gravity[3];
  double dt, c;
  int32_t parent[DDP_MAXJ];
  int32_t jtype[DDP_MAXJ];
  double axis[DDP_MAXJ][3];
  double Rp[DDP_MAXJ][9];
  double pp[DDP_MAXJ][3];
  double I6[DDP_MAXJ][21];  // body spatial inertia, packed lower triangle (row-major: (r,c), c<=r at r(r+1)/2+c)
  // small-state traversal tables (ctx.hip:build_slot_tables): a joint's running sum lives in one of a few slots
  int32_t has_child[DDP_MAXJ];      // joint has at least one child
  int32_t first_contrib[DDP_MAXJ];  // joint is the largest-index child of its parent (contributes first, leaf->root)
  int32_t last_child[DDP_MAXJ];     // joint is the largest-index child of its parent (read last, root->leaf)
  int32_t slot_up[DDP_MAXJ];        // slot of the joint's accumulator in the leaf->root pass (-1: leaf)
  int32_t slot_down[DDP_MAXJ];      // slot of the joint's value in the root->leaf pass (-1: leaf)
  int32_t n_slots;
  // level schedule for the wave-cooperative traversals (rbd.h: aba_tree_coop): joints sorted by tree depth
  int32_t n_levels, max_level_width;
  int32_t lvl_start[DDP_MAXJ + 1];  // joints of level L are lvl_joint[lvl_start[L] .. lvl_start[L+1])
  int32_t lvl_joint[DDP_MAXJ];
  int32_t child_start[DDP_MAXJ + 1]; // children of joint i (descending index) are child_list[child_start[i] .. child_start[i+1])
  int32_t child_list[DDP_MAXJ];
  // constraint chain
  int32_t eq_kind, eq_advance, frame_joint, first_order_fd, fd_mode, pad_;
  double frame_off[3];
};

// Everything a kernel needs to find a block of a flat sequence.
struct Dims {
  int64_t T, n, m, nx, nv, batch;
  int64_t Etot;   // sum_t ne[t]
  int64_t emax;
};

struct SeqBuf {
  double* ptr = nullptr;   // [batch][size]
  int64_t size = 0;        // per-instance element count
};

struct BwdJob {
  int32_t kind;   // 0: x-columns (Q_xx, Q_ux, Q_x), 1: u-columns (Q_uu, Q_u)
  int32_t c0, cn;
  int32_t pad_;
};

struct ProfSlot {
  std::vector<hipEvent_t> starts, stops;
  size_t used = 0;
  double total_ms = 0;
  int64_t launches = 0;
};

struct ddp_hip_ctx {
  int device = 0;
  uint32_t flags = 0;
  hipStream_t stream = nullptr;
  Dims d{};
  std::vector<int64_t> ne_h, Epre_h;   // [T], [T+1]
  int64_t* ne_d = nullptr;             // [T]
  int64_t* Epre_d = nullptr;           // [T+1] prefix sums of ne
  double* target_d = nullptr;          // [Etot]
  DevModel model_h{};
  DevModel* model_d = nullptr;
  SeqBuf seq[DDP_HIP_SEQ_COUNT];

  // backward workspace, per instance
  double* ws_V = nullptr;      // [batch][n + n*n]          V_x | V_xx
  double* ws_Q = nullptr;      // [batch][n + m + n*n + m*n + m*m]   Q_x | Q_u | Q_xx | Q_ux | Q_uu
  double* ws_D = nullptr;      // [batch][n*n + m*n + m*m]   dense f^T V_xx f of the next step (split kernels)
  double* reg_d = nullptr;     // [batch]
  double* mu_d = nullptr;      // [batch]
  int32_t* status_d = nullptr; // [batch] 0 active, 1 failed this attempt, 2 done
  int64_t* restarts_d = nullptr;
  BwdJob* jobs_d = nullptr;
  BwdJob* jobs_half_d = nullptr;  // K3h's job list (bwd_split.h: bwd_contract_half)
  int32_t njobs_half = 0;
  bool tensor_tops_sparse = false; // the tensors' configuration rows are as the static mode-2 stencil leaves them: zeros but the two entries per column
  BwdJob* jobs_sym_d = nullptr;   // K3's job list for symmetric tensors (bwd_split.h, job kind 2)
  bool fxx_mirror_pending = false; // the static stencil left f_xx(:, q_i, v_c) out (lin.hip: lin_materialize_fxx forms it on demand)
  bool tensor_tops_zero = false;   // rows k < nv of every column of FXX / FUX / FUU hold zeros (what LinParams::skip_top relies on)
  bool tensors_sym = false;       // FXX / FUU hold what this context's own mode-2 (or tensor-free: zero) linearisation wrote: symmetric bit for bit
  int32_t njobs = 0;
  int32_t cbx = 0, cbu = 0;
  // the batch is swept in groups on their own streams: K3 of one group overlaps K4 of the others (bwd.hip)
  int32_t bwd_groups = 1;
  hipStream_t bwd_stream[8] = {};
  hipEvent_t bwd_ev_start = nullptr, bwd_ev_done[8] = {};
  hipStream_t bwd_side = nullptr;          // K5 of a step runs here, beside K3 on the main stream (bwd.hip: enqueue_sweep_v2)
  hipEvent_t bwd_ev_fork = nullptr, bwd_ev_join = nullptr;
  int bwd_fork = 0;
  // the sweep as an instantiated hipGraph (600 launches per group and sweep otherwise pay the enqueue cost every time);
  // one per state of the kernel arguments (the X buffers trade places at every swap_traj)
  struct BwdGraph { const void* key_x = nullptr; uint64_t key_misc = 0; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; };
  BwdGraph bwd_graph[4];
  int bwd_graph_next = 0;
  int bwd_use_graph = 1;
  size_t bwd_k3_lds_pad = 0;

  // forward workspace
  double* fw_x = nullptr;      // [batch][n_alpha_max][(T+1)*nx]
  double* fw_u = nullptr;      // [batch][n_alpha_max][T*m]
  double* fw_dcost = nullptr;  // [batch][n_alpha_max]
  double* fw_cost = nullptr;   // [batch][n_alpha_max][T+1] candidates' cost terms (constrained problems on the latency path)
  double* fw_cost_old = nullptr; // [batch]
  double* step_d = nullptr;    // [batch]
  int32_t* fw_state_d = nullptr; // [batch] 0 searching, 1 accepted, 2 floor hit
  double* fw_dcost_acc_d = nullptr; // [batch]
  int32_t n_alpha_max = 8;
  double* pick_pair_d = nullptr;  // best-cost pick of a single-rank run (comm.cpp: ddp_hip_shard_pick)

  // linearize workspace
  double* eq_ws = nullptr;     // constraint-chain workspace (large models)
  double* lin_ws = nullptr;
  size_t lin_ws_bytes = 0;
  int32_t lin_ncfg = 0, lin_nvcfg = 0;   // q- / v-cache entries per (instance, t)
  double* lin_qws = nullptr;   // configuration-level workspace of the static path, lin_qws_bt (instance, t) pairs at a time
  int64_t lin_qws_bt = 0;
  double* lin_qws2 = nullptr;  // second workspace + stream + events: the two configuration-level kernels of consecutive slices overlap
  hipStream_t lin_stream2 = nullptr;
  hipEvent_t lin_ev_up[2] = {nullptr, nullptr}, lin_ev_dn[2] = {nullptr, nullptr};
  double* ana_T = nullptr;     // analytic-derivative workspace (lin_analytic.hip): T = [dtau/dq | dtau/dv] per evaluation of a slice
  double* ana_M = nullptr;     // ... and M / M^-1 per configuration of a slice
  double* ana_F = nullptr;     // ... and the v rows of f_x at the perturbed points (mode-1 constraint tensors)
  bool ana_A_fresh = false;    // ana_A was formed by stage 0 of the linearisation call in progress
  bool ana_M0_fresh = false;   // ana_M0 was written by stage 0 of the linearisation call in progress
  bool fuu_zero = false;       // analytic mode 1: F_UU holds the exact zeros lin_analytic.hip left there (cleared by every other writer)
  double* ana_M0 = nullptr;    // [B T][nv][nv] M^-1 at the trajectory points (fused analytic path, mode 1)
  bool ana_split = false;      // development: three-kernel analytic path (DDP_HIP_ANA_SPLIT)
  double* ana_A = nullptr;     // [B T][2nv][nv] accelerations of the mode-1 perturbed points (static first-order kernels, level 6)
  int64_t ana_nbt = 0;         // (instance, t) pairs per slice
  int lin_static = 0;          // id of the compiled-in topology the model's tree matches (lin_static.hip), 0 = none

  // per-instance activity (ddp_hip_set_active): an inactive instance is frozen -- the sweeps skip it and swap_traj
  // keeps its trajectory (solve<M> returns an instance at its first optimum, ddp.hpp:799-800)
  std::vector<int32_t> active_h;   // [batch], 1 = active
  bool all_active = true;

  bool async_mode = false;     // ddp_hip_set_async: entry points that hand nothing back to the host do not wait for the stream
  uint32_t profile_mask = 0;   // bit (1 + kernel_id): that kernel class is bracketed by HIP events
  ProfSlot prof[DDP_HIP_K_COUNT];
};

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e__ = (expr);                            \
    if (e__ != hipSuccess) { (void)hipGetLastError(); return DDP_HIP_E_HIP; } \
  } while (0)

// end of an entry point that returns nothing to the host: wait for the stream unless the context is in asynchronous mode
#define END_SYNC(ctx) do { if (!(ctx)->async_mode) HIP_TRY(hipStreamSynchronize((ctx)->stream)); } while (0)

// profile helpers (ctx.hip)
void prof_begin(ddp_hip_ctx* ctx, int kid, hipStream_t stream = nullptr);   // stream: the one the kernel is launched on (default: the context's)
void prof_end(ddp_hip_ctx* ctx, int kid, hipStream_t stream = nullptr);

// per-op launchers implemented in their own translation units
int bwd_setup(ddp_hip_ctx* ctx);
void bwd_teardown(ddp_hip_ctx* ctx);
int fwd_setup(ddp_hip_ctx* ctx);
bool fwd_lat_supported(const ddp_hip_ctx* ctx);   // the latency kernels of the forward sweep apply (tree, no constraints, Talos size)
void fwd_teardown(ddp_hip_ctx* ctx);
int lin_setup(ddp_hip_ctx* ctx);
int lin_materialize_fxx(ddp_hip_ctx* ctx);   // FXX complete for readers outside the symmetric sweep (lin.hip)
void lin_teardown(ddp_hip_ctx* ctx);

// best-cost pick, device side (pick.hip): {cost, global index} of the local best / of G gathered pairs
int pick_local_launch(ddp_hip_ctx* ctx, int64_t rank, int64_t nranks, double* out_pair, hipStream_t stream);
int pick_final_launch(const double* pairs, int G, double* out_pair, hipStream_t stream);

static inline int64_t seq_block_offset_regular(int64_t t, int64_t stride) { return t * stride; }
