/*
 * ddp_oracle.h -- CPU restatement (plain C99, double) of the reference's DDP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the CPU baseline.  The product path (ddp_pinocchio_amd/csrc -> libddp_hip.so)
 * never links or calls it.
 *
 * Parity status.  The reference (s-elkazdadi/ddp-pinocchio) cannot be built here: Eigen 3.3.7,
 * Boost, fmt, doctest and Pinocchio are absent and its build fetches them over the network.
 * What the reference's own tests pin is the flat indexing (test/indexing.cpp, test/mat_seq.cpp);
 * those known answers are checked in tests/test_indexing.py.  The DDP sweep itself has no golden
 * numbers in the reference ("parity unpinned" by the reference); this restatement is pinned
 * instead by an independent numpy / mpmath restatement (oracle/np_oracle.py, fixtures under
 * tests/golden/) and by closed forms (pendulum) -- see DESIGN.md.
 * The rigid-body arithmetic (ABA) lives in Pinocchio (third party, version unpinned by the
 * reference's CMakeLists.txt:81-87); here it is restated from the published algorithm
 * (Featherstone, Rigid Body Dynamics Algorithms, Table 7.1) and checked by RNEA/CRBA identities.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef DDP_ORACLE_H
#define DDP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODEL_PENDULUM = 0, ORC_MODEL_TREE = 1 };
enum { ORC_EQ_NONE = 0, ORC_EQ_CONFIG = 1, ORC_EQ_FRAME = 2 };
enum { ORC_JOINT_REVOLUTE = 0, ORC_JOINT_PRISMATIC = 1, ORC_JOINT_FREEFLYER = 2 /* joint 0 only: SE(3), q = [p, quat xyzw], v = [lin, ang] body frame */ };

/* Model concept: include/ddp/pinocchio_model.hpp:77-186, include/ddp/pendulum_model.hpp:10-133.
 * Only vector-space configurations (nq == nv, all joints 1-DoF) are restated: the reference's
 * FD mode 1 asserts exactly that (include/ddp/problem.hpp:78-81). */
typedef struct orc_model {
  int32_t kind;
  int32_t nv;              /* = nq */
  /* pendulum (pendulum_model.hpp:24-26): g = 9.81 */
  double mass, length;
  /* tree of 1-DoF joints, joints sorted so that parent[i] < i; parent = -1 is the world */
  const int32_t* parent;   /* [nv] */
  const int32_t* jtype;    /* [nv] */
  const double* axis;      /* [nv*3] unit axis in the joint frame */
  const double* Rp;        /* [nv*9] row-major rotation: parent-frame coords = Rp * joint-frame coords */
  const double* pp;        /* [nv*3] joint origin in the parent frame */
  const double* mass_j;    /* [nv]   */
  const double* com;       /* [nv*3] centre of mass in the body (joint) frame */
  const double* Ic;        /* [nv*9] rotational inertia about the com, body frame */
  double gravity[3];       /* world-frame gravitational acceleration, e.g. (0,0,-9.81) */
} orc_model;

/* problem_t + dynamics_t + constraint chain: include/ddp/problem.hpp:343-525,527-870,872-1150 */
typedef struct orc_problem {
  orc_model model;
  double dt;               /* dynamics_t::dt,  problem.hpp:522 */
  double c;                /* problem_t::c,    problem.hpp:1147 (cost l = c/2 |u|^2) */
  int64_t T;               /* horizon: index_begin = 0, index_end = T */
  int32_t eq_kind;         /* ORC_EQ_* */
  int32_t eq_advance;      /* number of constraint_advance_time_t wrappers (reference drivers: 2) */
  const int64_t* ne;       /* [T] eq rows at solver time t (after the time shift) */
  const double* eq_target; /* flat, concatenated over t: ne[t] doubles each */
  int32_t frame_joint;     /* ORC_EQ_FRAME: joint the frame is attached to */
  double frame_off[3];     /* frame origin in that joint's frame */
  int32_t first_order_fd;  /* 0: analytic first_order_deriv (problem.hpp:463-503); 1: forward FD, eps = sqrt(eps_mach) */
  int32_t fd_mode;         /* second order: 0 none (zero tensors), 1 (problem.hpp:67-150), 2 (problem.hpp:152-298) */
} orc_problem;

/* derivative_storage_t: include/ddp/ddp.hpp:52-245; flat layouts per detail/mat_seq.hpp:61-73 and
 * detail/tensor.hpp:141-147.  All blocks column-major; 1 x k blocks are k contiguous doubles. */
typedef struct orc_derivs {
  double *lfx, *lfxx;
  double *lx, *lu, *lxx, *lux, *luu;
  double *f_val, *fx, *fu, *fxx, *fux, *fuu;
  double *eq_val, *eq_x, *eq_u, *eq_xx, *eq_ux, *eq_uu;
} orc_derivs;

/* affine_vector_function_seq_t: include/ddp/detail/mat_seq_common.hpp:12-177 */
typedef struct orc_affine {
  double *origin, *val, *jac;
} orc_affine;

int64_t orc_nx(const orc_problem* p);
int64_t orc_ndx(const orc_problem* p);
int64_t orc_nu(const orc_problem* p);
int64_t orc_ne_total(const orc_problem* p);

/* ---- model level ------------------------------------------------------------------------- */
void orc_aba(const orc_model* m, const double* q, const double* v, const double* tau, double* qdd);
void orc_rnea(const orc_model* m, const double* q, const double* v, const double* a, double* tau);
void orc_crba(const orc_model* m, const double* q, double* M /* nv x nv col-major */);
/* partials of tau = RNEA(q, v, a) wrt q and v, and the joint-space inertia matrix (nv x nv col-major each) */
void orc_rnea_derivatives(const orc_model* m, const double* q, const double* v, const double* a,
                          double* dtau_dq, double* dtau_dv, double* M);
/* model_t::d_dynamics_aba (pinocchio_model.ipp:359-400): partials of qdd = ABA(q, v, tau) wrt q, v, tau */
void orc_aba_derivatives(const orc_model* m, const double* q, const double* v, const double* tau,
                         double* dq, double* dv, double* dtau);
void orc_frame_position(const orc_model* m, int32_t joint, const double* off, const double* q, double* p3);
/* top three rows of the WORLD-frame jacobian as the reference takes them
 * (pinocchio_model.ipp:458-461); world_aligned != 0 gives d(position)/dq instead */
void orc_frame_jacobian(const orc_model* m, int32_t joint, const double* off, const double* q,
                        int world_aligned, double* J /* 3 x nv col-major */);

/* ---- Lie-group configurations (pinocchio_model.ipp:222-321); nq = nv + 1 with a free-flyer root, else nq = nv ---- */
int32_t orc_model_nq(const orc_model* m);
/* test hook: {sin t/t, (1-cos t)/t^2, (t-sin t)/t^3, (1-(t/2)cot(t/2))/t^2, (1-t^2/2-cos t)/t^4, (t-sin t-t^3/6)/t^5} at t^2 */
void orc_so3_coeffs(double t2, double* out6);
void orc_integrate(const orc_model* m, const double* q, const double* v, double* out_q);
void orc_difference(const orc_model* m, const double* q_start, const double* q_finish, double* out_v);
void orc_d_integrate_dq(const orc_model* m, const double* q, const double* v, double* out /* nv x nv col-major */);
void orc_d_integrate_dv(const orc_model* m, const double* q, const double* v, double* out);
void orc_d_difference_dq_start(const orc_model* m, const double* q_start, const double* q_finish, double* out);
void orc_d_difference_dq_finish(const orc_model* m, const double* q_start, const double* q_finish, double* out);

/* ---- dynamics / constraints -------------------------------------------------------------- */
void orc_eval_f(const orc_problem* p, const double* x, const double* u, double* x_out);
void orc_first_order_f(const orc_problem* p, const double* x, const double* u,
                       double* fx, double* fu, double* f);
void orc_eval_eq(const orc_problem* p, int64_t t, const double* x, const double* u, double* out);
void orc_cost_seq_aug(const orc_problem* p, const double* xs, const double* us,
                      const orc_affine* mults, double mu, double* costs /* T+1 */);
void orc_compute_derivatives(const orc_problem* p, const double* xs, const double* us, orc_derivs* d);
void orc_rollout(const orc_problem* p, const double* x0, const double* us, double* xs);

/* ---- solver ------------------------------------------------------------------------------ */
/* backward_pass<primal_dual_affine_multipliers>: include/ddp/ddp_bwd.ipp:9-155.
 * reg/mu are in-out.  Vx_trace (T*n) / Vxx_trace (T*n*n), if non-null, receive V after each step t.
 * heap_like != 0 allocates every per-step temporary like the reference does (ddp_bwd.ipp:27-83).
 * max_restarts bounds the while(!success) loop (the reference has no bound); returns the number
 * of restarts, or -1 if the bound was hit. */
int64_t orc_backward(int64_t T, int64_t n, int64_t m, int64_t nx, const int64_t* ne,
                     const orc_derivs* d, const double* xs, const orc_affine* mults,
                     double* reg, double* mu, orc_affine* fb,
                     double* Vx_trace, double* Vxx_trace, int heap_like, int64_t max_restarts);

/* forward_pass: include/ddp/ddp_fwd.ipp:9-67 (sequential step halving). xs_new[0..nx) must be preset.
 * Returns the step (even when the line search fails, step < 1e-10). n_evals counts rollouts. */
double orc_forward(const orc_problem* p, double* xs_new, double* us_new,
                   const double* xs_old, const double* us_old,
                   const orc_affine* mults, const orc_affine* fb, double mu, int64_t* n_evals);
/* one closed-loop rollout at a fixed step + its summed cost difference (ddp_fwd.ipp:39-56) */
double orc_forward_alpha(const orc_problem* p, double step, double* xs_new, double* us_new,
                         const double* xs_old, const double* us_old,
                         const orc_affine* mults, const orc_affine* fb, double mu);

/* affine_vector_function_seq_t::update_origin: mat_seq_common.hpp:62-89. rows[t] = rows of block t */
void orc_update_origin(const orc_problem* p, orc_affine* a, const int64_t* rows, const double* xs_new);
double orc_optimality_constr(const orc_problem* p, const orc_derivs* d);          /* ddp.hpp:516-523 */
double orc_optimality_obj(const orc_problem* p, const double* xs, const orc_affine* mults,
                          double mu, const orc_derivs* d);                        /* ddp.hpp:576-627 */

typedef struct orc_solve_log {
  int64_t iterations;
  int64_t result;          /* 0 max_iterations reached, 1 optimum_attained */
  double mu, reg, w, n, last_step, opt_obj, opt_constr;
} orc_solve_log;

/* solve<primal_dual_affine_multipliers>: include/ddp/ddp.hpp:745-842.
 * xs/us: in = initial trajectory, out = final.  mult_jac_seed: initial multiplier jacobians
 * (the reference draws them with setRandom(), ddp.hpp:762; here they are an input). */
void orc_solve(const orc_problem* p, int64_t max_iterations, double threshold, double mu, double reg,
               double w, double n, const double* mult_jac_seed, double* xs, double* us,
               orc_affine* fb_out, orc_solve_log* log);

#ifdef __cplusplus
}
#endif
#endif
