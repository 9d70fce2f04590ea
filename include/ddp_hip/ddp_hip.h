/*
 * ddp_hip.h -- C-ABI of the MI355X-native DDP sweep library (libddp_hip.so).
 *
 * This is the drop-in boundary of the build.  The reference (s-elkazdadi/ddp-pinocchio) has no
 * FFI: its seams are C++ template/ODR seams.  Each entry point below replaces one
 * reference interface (cited file:line, relative to the reference root) and is what a host-side
 * binding (the Eigen adapter of INTEGRATION.md, a ctypes stub, ...) would bind:
 *
 *   reference interface                                            entry point
 *   -------------------------------------------------------------  ---------------------------
 *   ddp_solver_t ctor + uninit_derivative_storage ddp.hpp:430-514   ddp_hip_create / _destroy
 *   model_t ctor (pinocchio_model.ipp:119-160)                      ddp_hip_create (model table)
 *   mat_seq_t::data() views  detail/mat_seq.hpp:43-44               ddp_hip_upload / _download / _device_ptr
 *   make_trajectory          ddp.hpp:392-415                        ddp_hip_rollout
 *   problem_t::compute_derivatives  problem.hpp:956-998             ddp_hip_linearize
 *   backward_pass<M>         ddp_bwd.ipp:9-155 (decl ddp.hpp:845)   ddp_hip_backward
 *   forward_pass<M>          ddp_fwd.ipp:9-67  (decl ddp.hpp:855)   ddp_hip_forward
 *   cost_seq_aug             ddp.hpp:699-735                        ddp_hip_cost_seq_aug
 *   swap(traj, new_traj)     ddp.hpp:826                            ddp_hip_swap_traj
 *   update_origin            detail/mat_seq_common.hpp:62-89        ddp_hip_update_origin
 *   optimality_obj / _constr ddp.hpp:576-627, 516-523               ddp_hip_optimality
 *   multiplier update        ddp.hpp:680-688 (in update_derivatives) ddp_hip_update_multipliers
 *   solve<M>                 ddp.hpp:745-842                        ddp_hip_solve (+ ddp_hip_set_active)
 *   (new: multi-seed shard)                           ddp_hip_comm_* / ddp_hip_shard_best
 *
 * Conventions
 *  - All scalars are IEEE double; all dimensions and indices int64_t (utils.hpp:117 index_t).
 *  - A context owns `batch` independent problem instances (same model and horizon, different
 *    trajectories / derivatives); every sequence is resident in HBM as [batch][flat], where `flat`
 *    is exactly the reference's flat layout: block t at sum_{s<t} rows(s)*cols(s), column-major
 *    (detail/mat_seq.hpp:61-73); tensors (i=out, j=left, k=right) at i + j*O + k*O*L
 *    (detail/tensor.hpp:141-147).  ddp_hip_seq_size() returns the per-instance element count.
 *  - Return value: 0 ok, >0 numerical event, <0 usage / HIP error.  Nothing throws, nothing
 *    terminates.  One context = one HIP stream; contexts are independent; a single context is not
 *    re-entrant.  There is NO CPU fallback: without a HIP device ddp_hip_create fails with
 *    DDP_HIP_E_NODEVICE.
 */
#ifndef DDP_HIP_H
#define DDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDP_HIP_ABI_VERSION 3

enum {
  DDP_HIP_OK = 0,
  DDP_HIP_EV_LLT_RESTART = 1,       /* at least one instance restarted its backward sweep (ddp_bwd.ipp:105-132) */
  DDP_HIP_EV_LINESEARCH_FLOOR = 2,  /* at least one instance hit step < 1e-10 (ddp_fwd.ipp:35-37) */
  DDP_HIP_E_ARG = -1,
  DDP_HIP_E_HIP = -2,
  DDP_HIP_E_NODEVICE = -3,
  DDP_HIP_E_UNSUPPORTED = -4,
  DDP_HIP_E_MAX_RESTARTS = -5,
  DDP_HIP_E_COMM = -6
};

enum { DDP_HIP_MODEL_PENDULUM = 0, DDP_HIP_MODEL_TREE = 1 };
enum { DDP_HIP_EQ_NONE = 0, DDP_HIP_EQ_CONFIG = 1, DDP_HIP_EQ_FRAME = 2 };
/* FREEFLYER: joint 0 only (parent -1): an SE(3) joint, q = [p(3), quaternion x y z w], v = [linear(3), angular(3)] in the
 * body frame (Pinocchio's JointModelFreeFlyer); the model then has nq = nv + 1 and nv - 5 joints */
enum { DDP_HIP_JOINT_REVOLUTE = 0, DDP_HIP_JOINT_PRISMATIC = 1, DDP_HIP_JOINT_FREEFLYER = 2 };

#define DDP_HIP_MAX_JOINTS 64

/* Model concept (pinocchio_model.hpp:77-186, pendulum_model.hpp:10-133): a tree of 1-DoF joints (nq == nv, the case FD
 * mode 1 requires: problem.hpp:78-81), optionally hanging from a free-flyer root (a Lie-group configuration, nq = nv + 1:
 * pinocchio_model.ipp:222-321), or the closed-form pendulum.  The per-joint arrays have nv entries, nv - 5 with a free flyer. */
typedef struct ddp_hip_model {
  int32_t kind;
  int32_t nv;                    /* velocity / tangent dimension */
  double mass, length;           /* pendulum: pendulum_model.hpp:24-26 (g = 9.81) */
  const int32_t* parent;         /* [nv], parent[i] < i, -1 = world */
  const int32_t* jtype;          /* [nv] DDP_HIP_JOINT_* */
  const double* axis;            /* [nv*3] unit joint axis, joint frame */
  const double* Rp;              /* [nv*9] row-major: parent coords = Rp * joint coords */
  const double* pp;              /* [nv*3] joint origin in the parent frame */
  const double* mass_j;          /* [nv] */
  const double* com;             /* [nv*3] */
  const double* Ic;              /* [nv*9] rotational inertia about the com */
  double gravity[3];
} ddp_hip_model;

/* problem_t / dynamics_t / constraint chain (problem.hpp:343-525, 527-870, 872-1150) */
typedef struct ddp_hip_problem {
  ddp_hip_model model;
  double dt;                     /* dynamics_t::dt  problem.hpp:522 */
  double c;                      /* problem_t::c    problem.hpp:1147 */
  int64_t T;                     /* horizon; index_begin = 0, index_end = T */
  int64_t batch;                 /* independent instances resident in this context */
  int32_t eq_kind;               /* DDP_HIP_EQ_* */
  int32_t eq_advance;            /* constraint_advance_time_t wrappers (reference drivers: 2) */
  const int64_t* ne;             /* [T] eq rows at solver time t; NULL = no constraints */
  const double* eq_target;       /* concatenated over t, ne[t] doubles each (shared by the batch) */
  int32_t frame_joint;
  double frame_off[3];
  int32_t first_order_fd;        /* 0 analytic (pendulum only), 1 forward FD with eps = sqrt(DBL_EPSILON) */
  int32_t fd_mode;               /* second order: 0 none (tensors zero), 1 problem.hpp:67-150, 2 problem.hpp:152-298 */
} ddp_hip_problem;

/* resident sequences (derivative_storage_t ddp.hpp:52-245, trajectory_t trajectory.hpp:9-113,
 * affine_vector_function_seq_t mat_seq_common.hpp:12-177) */
enum ddp_hip_seq {
  DDP_HIP_SEQ_X = 0, DDP_HIP_SEQ_U, DDP_HIP_SEQ_X_NEW, DDP_HIP_SEQ_U_NEW,
  DDP_HIP_SEQ_LFX, DDP_HIP_SEQ_LFXX,
  DDP_HIP_SEQ_LX, DDP_HIP_SEQ_LU, DDP_HIP_SEQ_LXX, DDP_HIP_SEQ_LUX, DDP_HIP_SEQ_LUU,
  DDP_HIP_SEQ_F_VAL, DDP_HIP_SEQ_FX, DDP_HIP_SEQ_FU, DDP_HIP_SEQ_FXX, DDP_HIP_SEQ_FUX, DDP_HIP_SEQ_FUU,
  DDP_HIP_SEQ_EQ_VAL, DDP_HIP_SEQ_EQ_X, DDP_HIP_SEQ_EQ_U, DDP_HIP_SEQ_EQ_XX, DDP_HIP_SEQ_EQ_UX, DDP_HIP_SEQ_EQ_UU,
  DDP_HIP_SEQ_MULT_ORIGIN, DDP_HIP_SEQ_MULT_VAL, DDP_HIP_SEQ_MULT_JAC,
  DDP_HIP_SEQ_FB_ORIGIN, DDP_HIP_SEQ_FB_VAL, DDP_HIP_SEQ_FB_JAC,
  DDP_HIP_SEQ_VX_TRACE, DDP_HIP_SEQ_VXX_TRACE,   /* V_x / V_xx after every step (parity only) */
  DDP_HIP_SEQ_COSTS_OLD, DDP_HIP_SEQ_COSTS_NEW,  /* T+1 doubles each (cost_seq_aug) */
  DDP_HIP_SEQ_COUNT
};

typedef struct ddp_hip_ctx ddp_hip_ctx;

/* create flags */
#define DDP_HIP_FLAG_NO_TENSORS 1u   /* do not allocate fxx/fux/fuu/eq_xx/eq_ux/eq_uu (Gauss-Newton sweeps only) */
#define DDP_HIP_FLAG_TRACE 2u        /* allocate the V_x / V_xx trace sequences */

int ddp_hip_abi_version(void);
const char* ddp_hip_strerror(int code);
int ddp_hip_device_count(void);

int ddp_hip_create(const ddp_hip_problem* prob, int device, uint32_t flags, ddp_hip_ctx** out);
int ddp_hip_destroy(ddp_hip_ctx* ctx);
/* the context's HIP stream (hipStream_t as void*) */
void* ddp_hip_stream(ddp_hip_ctx* ctx);
int ddp_hip_synchronize(ddp_hip_ctx* ctx);
/* Asynchronous mode (off by default).  On: the entry points that hand nothing back to the host -- ddp_hip_linearize[_stages],
 * ddp_hip_update_origin, ddp_hip_update_multipliers, ddp_hip_swap_traj -- enqueue their work on the context's stream and return
 * without waiting; the entry points that return values (ddp_hip_optimality, ddp_hip_backward, ddp_hip_forward, downloads ...)
 * synchronise as always, and everything is stream-ordered.  ddp_hip_solve runs its loop in this mode: per iteration the host
 * waits three times -- for the stopping test's two scalars, for the sweep's restart status, for the line search's accept state
 * -- instead of at every call.  Switching it off waits for the stream. */
int ddp_hip_set_async(ddp_hip_ctx* ctx, int on);

int64_t ddp_hip_batch(const ddp_hip_ctx* ctx);                       /* instances resident in the context */
int64_t ddp_hip_seq_size(const ddp_hip_ctx* ctx, int seq);           /* elements per instance */
double* ddp_hip_device_ptr(ddp_hip_ctx* ctx, int seq);               /* [batch][seq_size], device memory */
int ddp_hip_upload(ddp_hip_ctx* ctx, int seq, const double* host, int64_t first_instance, int64_t n_instances);
int ddp_hip_download(ddp_hip_ctx* ctx, int seq, double* host, int64_t first_instance, int64_t n_instances);
int ddp_hip_fill(ddp_hip_ctx* ctx, int seq, double value);

/* make_trajectory (ddp.hpp:392-415): X[0] and U given -> X[1..T] */
int ddp_hip_rollout(ddp_hip_ctx* ctx);
/* compute_derivatives (problem.hpp:956-998) along (X, U) -> all derivative sequences */
int ddp_hip_linearize(ddp_hip_ctx* ctx);
/* the same, stage by stage (cost terms :982-987 | first order f :463-503 | second order f :50-341 |
 * constraint chain :527-870); the later stages read what the earlier ones left resident */
#define DDP_HIP_LIN_COST 1u
#define DDP_HIP_LIN_FIRST 2u
#define DDP_HIP_LIN_SECOND 4u
#define DDP_HIP_LIN_EQ 8u
int ddp_hip_linearize_stages(ddp_hip_ctx* ctx, uint32_t stages);

/* backward_pass<primal_dual_affine_multipliers> (ddp_bwd.ipp:9-155).
 * reg_io / mu_io: host arrays [batch], in-out (ddp_bwd.ipp:106-110,154); restarts_out: host [batch] or NULL.
 * max_restarts bounds the reference's unbounded while(!success). */
int ddp_hip_backward(ddp_hip_ctx* ctx, double* reg_io, double* mu_io, int64_t* restarts_out, int64_t max_restarts);

/* forward_pass (ddp_fwd.ipp:9-67) with the step halving evaluated n_alpha candidates at a time:
 * candidates 2^0 .. 2^-(n_alpha-1) roll out concurrently, the LARGEST accepted one is kept (the same
 * decision sequential halving makes); if none is accepted the next n_alpha candidates follow, until
 * step < 1e-10.  mu: host [batch]; step_out: host [batch]; dcost_out: host [batch] or NULL
 * (sum_t(cost_new - cost_old) of the returned step).  X_NEW[0] must hold x_0 (ddp.hpp:752).
 * n_alpha == 0: do_linesearch == false (ddp_fwd.ipp:61-63) -- the full step is rolled out once and taken whatever the
 * cost does (step_out = 1, X_NEW / U_NEW = that rollout). */
int ddp_hip_forward(ddp_hip_ctx* ctx, const double* mu, int32_t n_alpha, double* step_out, double* dcost_out);

/* cost_seq_aug (ddp.hpp:699-735) of (X,U) [which=0] or (X_NEW,U_NEW) [which=1] into COSTS_OLD / COSTS_NEW */
int ddp_hip_cost_seq_aug(ddp_hip_ctx* ctx, int which, const double* mu);
/* swap(traj, new_traj) (ddp.hpp:826) */
int ddp_hip_swap_traj(ddp_hip_ctx* ctx);

/* ---- outer augmented-Lagrangian loop (solve<M>, ddp.hpp:745-842): the parts of update_derivatives
 * (ddp.hpp:642-696) between compute_derivatives and backward_pass, on the resident sequences ---------- */
/* affine_vector_function_seq_t::update_origin (mat_seq_common.hpp:62-89) with x_new = X:
 * val += jac (X - origin); origin = X.  which = 0: the multipliers (MULT_*), 1: the control feedback (FB_*) */
int ddp_hip_update_origin(ddp_hip_ctx* ctx, int which);
/* optimality_obj (ddp.hpp:576-627) and optimality_constr (ddp.hpp:516-523) of (X, multipliers, derivatives);
 * mu, obj_out, constr_out: host [batch] */
int ddp_hip_optimality(ddp_hip_ctx* ctx, const double* mu, double* obj_out, double* constr_out);
/* p.val += mu (eq + eq_u k), p.jac += mu (eq_x + eq_u K) (ddp.hpp:680-688); mu: host [batch] */
int ddp_hip_update_multipliers(ddp_hip_ctx* ctx, const double* mu);

/* Per-instance activity.  solve<M> returns an instance the moment it reaches its optimum (ddp.hpp:799-800); in a batch
 * the others go on.  An inactive instance is frozen: ddp_hip_backward / ddp_hip_forward skip it (its reg / mu / step
 * entries are left as they are) and ddp_hip_swap_traj keeps its (X, U).  active: host [batch] of 0 / 1, NULL = all. */
int ddp_hip_set_active(ddp_hip_ctx* ctx, const int32_t* active);

/* solve<primal_dual_affine_multipliers> (ddp.hpp:745-842) for every instance of the context, each with the reference's
 * per-problem semantics: an instance stops at its first optimum (result 1, `iterations` = the iteration that found it)
 * or after max_iterations (result 0), whatever its batch-mates do.  In: X / U the initial trajectory, X_NEW / U_NEW a
 * clone of it (ddp.hpp:752), MULT_* the initial multipliers (val 0, jac the seed the reference draws with setRandom(),
 * origin = X: ddp.hpp:759-764).  Out: the final trajectory in X / U, the feedback in FB_*, log[batch]. */
typedef struct ddp_hip_solver_params {   /* solver_parameters_t, ddp.hpp:42-50 */
  int64_t max_iterations;
  double optimality_stopping_threshold;
  double mu, reg, w, n;
  int32_t n_alpha;        /* line-search candidates per round (1 = the reference's sequential halving) */
  int32_t pad_;
  int64_t max_restarts;   /* bound on the reference's unbounded while(!success) of backward_pass */
} ddp_hip_solver_params;
typedef struct ddp_hip_solve_log {
  int64_t iterations;
  int32_t result;         /* 0 max_iterations reached, 1 optimum attained */
  int32_t pad_;
  double mu, reg, w, n, last_step, opt_obj, opt_constr;
} ddp_hip_solve_log;
int ddp_hip_solve(ddp_hip_ctx* ctx, const ddp_hip_solver_params* params, ddp_hip_solve_log* log);

/* Which implementation each phase of this context runs (a model whose tree matches no compiled-in topology gets the
 * run-time-tree stencil kernels, several times slower: visible here instead of silently) */
typedef struct ddp_hip_info {
  int32_t device;
  int32_t lin_path;       /* 0 closed form (pendulum), 1 run-time-tree kernels (any topology), 2 static TopoTalos38, 3 static TopoChain6,
                           * >= 4 a generated static topology (tools/gen_topology.py -> csrc/topo_extra.h; shipped: 4 Arm7, 5 Biped12).
                           * With first_order == 2 and fd_mode 1 a static topology means: the forward dynamics of the perturbed points come
                           * from the static first-order kernels (one ABA per (instance, t) instead of 2 nv + 1) */
  int32_t first_order;    /* 0 analytic (pendulum_model.hpp:116-130), 1 forward FD (north star), 2 analytic ABA derivatives */
  int32_t bwd_path;       /* 0 run-time-shaped bwd_assemble / bwd_gains, 1 split K3 bwd_contract / K4 bwd_riccati */
  int32_t fwd_path;       /* 0 one lane per rollout, 1 latency path (two workgroups per instance; constrained problems: + parallel cost kernel) */
  int32_t has_tensors;
  int64_t hbm_bytes;      /* bytes of the resident sequences */
} ddp_hip_info;
int ddp_hip_ctx_info(const ddp_hip_ctx* ctx, ddp_hip_info* out);

/* ---- measurement ------------------------------------------------------------------------- */
enum ddp_hip_kernel_id {
  DDP_HIP_K_BWD_ASSEMBLE = 0,   /* Q assembly + tensor contraction (the HBM-bound kernel) */
  DDP_HIP_K_BWD_GAINS,          /* LLT, gains, V update */
  DDP_HIP_K_FWD_ROLLOUT,
  DDP_HIP_K_LIN_FIRST,
  DDP_HIP_K_LIN_SECOND,
  DDP_HIP_K_COUNT
};
/* when enabled, HIP events bracket every launch of the selected kernel classes on the context's stream.
 * on: 0 off; 1 every class; otherwise a bit mask, bit (1 + kernel_id) selects class kernel_id (an event pair costs
 * a few microseconds of stream time per launch: K4's 200 launches per sweep are worth leaving out of a timed run) */
int ddp_hip_profile_enable(ddp_hip_ctx* ctx, int on);
int ddp_hip_profile_reset(ddp_hip_ctx* ctx);
int ddp_hip_profile_get(ddp_hip_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);
/* algorithmic bytes of one backward sweep of ONE instance (SURVEY.md 8d formula B_bwd) */
int64_t ddp_hip_bwd_algorithmic_bytes(const ddp_hip_ctx* ctx);
/* bytes of f_xx / f_ux / f_uu the tensor contraction physically reads per (instance, step) with the tensors in their current
 * state: all of them (tensors from outside); the columns j >= c of slab c (this context's own mode-2 tensors: symmetric bit for
 * bit); or the lower halves of those columns plus the two non-zero entries of each upper half (the static stencil's tensors: the
 * configuration rows of f are affine, their second differences exact zeros) -- see csrc/bwd_split.h */
int64_t ddp_hip_bwd_stream_bytes(const ddp_hip_ctx* ctx);

/* ---- multi-GPU shard (new, no reference counterpart) ----------------------- */
#define DDP_HIP_COMM_ID_BYTES 128
typedef struct ddp_hip_comm ddp_hip_comm;
int ddp_hip_comm_unique_id(unsigned char id[DDP_HIP_COMM_ID_BYTES]);
int ddp_hip_comm_init(const unsigned char id[DDP_HIP_COMM_ID_BYTES], int rank, int nranks, int device, ddp_hip_comm** out);
int ddp_hip_comm_destroy(ddp_hip_comm* comm);
/* min over ranks of local_cost, and the smallest global index attaining it (RCCL has no MINLOC):
 * two 8-byte all-reduces over xGMI */
int ddp_hip_shard_best(ddp_hip_comm* comm, double local_cost, int64_t local_global_index,
                       double* best_cost, int64_t* best_global_index);
/* The pick on resident data, as ONE collective.  Ownership rule: global instance s lives on rank s mod G at local position
 * s div G.  Every rank forms the cost of the trajectory ddp_hip_forward just produced for each of its instances
 * (sum_t COSTS_OLD + the accepted step's cost difference), takes its local argmin on the device, one 16-byte ncclAllGather
 * of {cost, global index} over xGMI, argmin of the G pairs on the device, one 16-byte read-back.  comm == NULL: a single
 * rank (the same device work without the collective). */
int ddp_hip_shard_pick(ddp_hip_comm* comm, ddp_hip_ctx* ctx, double* best_cost, int64_t* best_global_index);
/* Optional: the winner's trajectory and gains (X, U, FB_ORIGIN, FB_VAL, FB_JAC of global instance best_global_index) from
 * its owner into local instance dst_local of every rank: one grouped ncclBroadcast between the resident sequences
 * (~4.9 MB at the Talos shape).  comm == NULL: a device copy. */
int ddp_hip_shard_broadcast(ddp_hip_comm* comm, ddp_hip_ctx* ctx, int64_t best_global_index, int64_t dst_local);

/* ---- the Model concept point by point (pinocchio_model.hpp:77-186), for a host-side model_t<double> (seam B2, see
 * adapters/pinocchio_double.cpp).  One configuration per call, evaluated on the device by the same rigid-body code the
 * batched entry points use: plumbing, not a hot path.  Matrices nv x nv column-major. ------------------------------ */
typedef struct ddp_hip_model_handle ddp_hip_model_handle;
int ddp_hip_model_create(const ddp_hip_model* model, int device, ddp_hip_model_handle** out);
int ddp_hip_model_destroy(ddp_hip_model_handle* h);
/* model_t::dynamics_aba, pinocchio_model.ipp:337-356 */
int ddp_hip_model_aba(ddp_hip_model_handle* h, const double* q, const double* v, const double* tau, double* qdd);
/* model_t::d_dynamics_aba, pinocchio_model.ipp:359-400 */
int ddp_hip_model_aba_derivatives(ddp_hip_model_handle* h, const double* q, const double* v, const double* tau,
                                  double* dq, double* dv, double* dtau);
/* model_t::frame_coordinates / d_frame_coordinates, pinocchio_model.ipp:418-462 (J: 3 x nv, may be NULL; the reference's
 * WORLD-frame rows) */
int ddp_hip_model_frame(ddp_hip_model_handle* h, int32_t joint, const double off[3], const double* q, double* p3, double* J);

/* ---- built-in seeded model tables (no URDF exists offline: SURVEY.md D4, 8d) -------------- */
/* ..._FF: the same robots on a free-flyer root instead of a fixed base / 3 prismatic + 3 revolute base joints */
enum { DDP_HIP_BUILTIN_PENDULUM = 0, DDP_HIP_BUILTIN_CHAIN6 = 1, DDP_HIP_BUILTIN_TREE38 = 2, DDP_HIP_BUILTIN_CHAIN6_FF = 3, DDP_HIP_BUILTIN_TREE38_FF = 4 };
/* fills caller-provided arrays (sized for DDP_HIP_MAX_JOINTS) and points `out` at them */
typedef struct ddp_hip_model_storage {
  int32_t parent[DDP_HIP_MAX_JOINTS];
  int32_t jtype[DDP_HIP_MAX_JOINTS];
  double axis[DDP_HIP_MAX_JOINTS * 3];
  double Rp[DDP_HIP_MAX_JOINTS * 9];
  double pp[DDP_HIP_MAX_JOINTS * 3];
  double mass_j[DDP_HIP_MAX_JOINTS];
  double com[DDP_HIP_MAX_JOINTS * 3];
  double Ic[DDP_HIP_MAX_JOINTS * 9];
} ddp_hip_model_storage;
int ddp_hip_builtin_model(int which, uint64_t seed, ddp_hip_model_storage* storage, ddp_hip_model* out);

#ifdef __cplusplus
}
#endif
#endif
