// ddp/ddp.hpp -- host-side mirror (dependency-free C++17, double only) of the reference's solver / problem / model
// interface for the hot path, on top of the C-ABI of libddp_hip.so.  Same public names and argument meaning as
//   ddp_solver_t            include/ddp/ddp.hpp:300-869      problem_t / dynamics_t   include/ddp/problem.hpp:343-525,872-1150
//   trajectory_t            include/ddp/trajectory.hpp:9-113 pendulum_model_t         include/ddp/pendulum_model.hpp:10-133
//   solver_parameters_t     include/ddp/ddp.hpp:42-50        pinocchio::model_t       include/ddp/pinocchio_model.hpp:15-186
// so that a driver written against the reference (test/pendulum_ddp.cpp, test/pinocchio_ddp.cpp) reads the same here.
// Differences forced by the boundary: matrices are flat column-major buffers instead of Eigen maps; the derivative
// storage stays resident in HBM (derivative_storage_t is a handle); errors come back as codes, never std::terminate.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ddp/detail/mat_seq.hpp"
#include "ddp/indexer.hpp"
#include "ddp_hip/ddp_hip.h"

namespace ddp {

enum struct method { primal, primal_dual_constant_multipliers, primal_dual_affine_multipliers };   // ddp.hpp:18-22
enum struct mult_update_attempt_result_e { no_update, update_success, update_failure, optimum_attained };   // :11-16

template <typename Scalar>
struct solver_parameters_t {   // ddp.hpp:42-50
  index_t max_iterations;
  Scalar optimality_stopping_threshold;
  Scalar mu, reg, w, n;
};

struct ddp_hip_error : std::runtime_error {
  int code;
  ddp_hip_error(int c, const char* where) : std::runtime_error(std::string(where) + ": " + ddp_hip_strerror(c)), code(c) {}
};
inline int check(int rc, const char* where) { if (rc < 0) throw ddp_hip_error(rc, where); return rc; }

// ---- models -------------------------------------------------------------------------------------------------------
template <typename T>
struct pendulum_model_t {   // pendulum_model.hpp:10-133
  using scalar_t = T;
  T m_mass, m_length;
  pendulum_model_t(T mass, T length) : m_mass(mass), m_length(length) {}
  index_t configuration_dim() const { return 1; }
  index_t tangent_dim() const { return 1; }
  void fill(ddp_hip_model& m, ddp_hip_model_storage& st) const {
    check(ddp_hip_builtin_model(DDP_HIP_BUILTIN_PENDULUM, 0, &st, &m), "builtin_model");
    m.mass = m_mass; m.length = m_length;
  }
  const char* model_name() const { return "pendulum"; }
};
namespace pinocchio {
// model_t: the reference wraps a Pinocchio model loaded from URDF (pinocchio_model.ipp:98-160); neither exists
// offline, so the built-in seeded tables stand in
template <typename T>
struct model_t {
  using scalar_t = T;
  int m_which; std::uint64_t m_seed; index_t m_nv;
  explicit model_t(int builtin, std::uint64_t seed = 1) : m_which(builtin), m_seed(seed) {
    ddp_hip_model m; ddp_hip_model_storage st;
    check(ddp_hip_builtin_model(builtin, seed, &st, &m), "builtin_model");
    m_nv = m.nv;
  }
  index_t configuration_dim() const { return m_nv; }
  index_t tangent_dim() const { return m_nv; }
  void fill(ddp_hip_model& m, ddp_hip_model_storage& st) const { check(ddp_hip_builtin_model(m_which, m_seed, &st, &m), "builtin_model"); }
  const char* model_name() const { return m_which == DDP_HIP_BUILTIN_CHAIN6 ? "chain6" : "tree38"; }
};
}  // namespace pinocchio

template <typename Model>
struct dynamics_t {   // problem.hpp:343-525
  using scalar_t = typename Model::scalar_t;
  using model_t = Model;
  Model const& m_model;
  scalar_t dt;
  bool second_order_finite_diff = true;
  index_t state_dim() const { return m_model.configuration_dim() + m_model.tangent_dim(); }
  index_t dstate_dim() const { return 2 * m_model.tangent_dim(); }
  index_t control_dim() const { return m_model.tangent_dim(); }
};

// equality constraint description: a config_constraint_t (problem.hpp:744-864) or spatial_constraint_t (:631-742)
// wrapped `advance` times in constraint_advance_time_t (:527-624); eq_idx is the already time-shifted indexer
struct constraint_t {
  int kind = DDP_HIP_EQ_NONE;
  int advance = 2;
  indexing::indexer_ptr m_eq_idx;          // rows(t) at solver time t
  std::vector<double> m_target;            // concatenated over t
  int frame_joint = 0;
  double frame_off[3] = {0, 0, 0};
  indexing::indexer_ptr eq_idx() const { return m_eq_idx; }
};

template <typename Dynamics>
struct problem_t {   // problem.hpp:872-1150
  using dynamics_t = Dynamics;
  using scalar_t = typename Dynamics::scalar_t;
  index_t m_begin, m_end;
  scalar_t c = 1e2;                        // problem.hpp:1147
  Dynamics m_dynamics;
  constraint_t m_constraint;
  index_t state_dim() const { return m_dynamics.state_dim(); }
  index_t dstate_dim() const { return m_dynamics.dstate_dim(); }
  const char* name() const { return m_dynamics.m_model.model_name(); }
};

// ---- containers ---------------------------------------------------------------------------------------------------
template <typename Scalar>
struct trajectory_t {   // trajectory.hpp:9-113: x[0..T] and u[0..T-1] as two flat sequences
  detail::matrix_seq::mat_seq_t<Scalar> m_state_data, m_control_data;
  trajectory_t(indexing::indexer_ptr x_idx, indexing::indexer_ptr u_idx) : m_state_data(std::move(x_idx)), m_control_data(std::move(u_idx)) {}
  index_t index_begin() const { return m_control_data.m_idx->index_begin(); }
  index_t index_end() const { return m_control_data.m_idx->index_end(); }
  Scalar* x(index_t t) { return m_state_data[t].data(); }
  Scalar* u(index_t t) { return m_control_data[t].data(); }
  Scalar* x_f() { return m_state_data[index_end()].data(); }
  trajectory_t clone() const { return *this; }
};

template <typename Scalar>
struct affine_vector_function_seq_t {   // mat_seq_common.hpp:12-177: value at x is val + jac (x - origin)
  detail::matrix_seq::mat_seq_t<Scalar> m_origin, m_val_data, m_jac_data;
  affine_vector_function_seq_t(indexing::indexer_ptr out_idx, index_t nx, index_t ndx)
      : m_origin(indexing::vec_regular_indexer(out_idx->index_begin(), out_idx->index_end(), nx)),
        m_val_data(out_idx),
        m_jac_data(indexing::outer_prod(out_idx, indexing::vec_regular_indexer(out_idx->index_begin(), out_idx->index_end(), ndx))) {}
};

// handle to the derivative sequences resident in HBM (derivative_storage_t, ddp.hpp:52-245)
struct derivative_storage_t {
  ddp_hip_ctx* ctx;
  std::vector<double> download(int seq) const {
    std::vector<double> v(static_cast<size_t>(ddp_hip_seq_size(ctx, seq)));
    check(ddp_hip_download(ctx, seq, v.data(), 0, 1), "download");
    return v;
  }
};

// ---- solver -------------------------------------------------------------------------------------------------------
template <typename Problem>
struct ddp_solver_t {   // ddp.hpp:300-869
  using problem_t = Problem;
  using scalar_t = typename Problem::scalar_t;
  using trajectory_t = ddp::trajectory_t<scalar_t>;
  using control_feedback_t = affine_vector_function_seq_t<scalar_t>;
  using multiplier_seq_t = affine_vector_function_seq_t<scalar_t>;
  template <method M> struct multiplier_sequence { struct type { multiplier_seq_t eq; }; };
  template <method M> struct backward_pass_result_t { control_feedback_t feedback; scalar_t mu; scalar_t reg; };

  Problem const& prob;
  indexing::indexer_ptr u_idx, eq_idx;
  std::vector<scalar_t> const& x_init;
  ddp_hip_ctx* ctx = nullptr;
  std::vector<std::int64_t> m_ne;
  index_t nx, ndx, nu, T, Etot = 0;

  ddp_solver_t(Problem const& p, indexing::indexer_ptr u, indexing::indexer_ptr eq, std::vector<scalar_t> const& x0, int device = 0)
      : prob(p), u_idx(std::move(u)), eq_idx(std::move(eq)), x_init(x0) {
    nx = prob.state_dim(); ndx = prob.dstate_dim(); nu = prob.m_dynamics.control_dim();
    T = index_end() - index_begin();
    if (index_begin() != 0) throw std::invalid_argument("index_begin must be 0 (ddp_bwd.ipp:149)");
    ddp_hip_problem hp{};
    ddp_hip_model_storage st;
    prob.m_dynamics.m_model.fill(hp.model, st);
    hp.dt = prob.m_dynamics.dt; hp.c = prob.c; hp.T = T; hp.batch = 1;
    hp.eq_kind = prob.m_constraint.kind; hp.eq_advance = prob.m_constraint.advance;
    m_ne.assign(static_cast<size_t>(T), 0);
    for (index_t t = 0; t < T; ++t) { m_ne[static_cast<size_t>(t)] = eq_idx ? eq_idx->rows(t) : 0; Etot += m_ne[static_cast<size_t>(t)]; }
    hp.ne = m_ne.data(); hp.eq_target = prob.m_constraint.m_target.data();
    hp.frame_joint = prob.m_constraint.frame_joint;
    for (int k = 0; k < 3; ++k) hp.frame_off[k] = prob.m_constraint.frame_off[k];
    hp.first_order_fd = hp.model.kind == DDP_HIP_MODEL_PENDULUM ? 0 : 1;
    hp.fd_mode = prob.m_dynamics.second_order_finite_diff ? 2 : (hp.first_order_fd ? 2 : 1);
    check(ddp_hip_create(&hp, device, 0, &ctx), "ddp_hip_create");
  }
  ~ddp_solver_t() { if (ctx) ddp_hip_destroy(ctx); }
  ddp_solver_t(ddp_solver_t const&) = delete;              // ddp.hpp:737-741
  ddp_solver_t& operator=(ddp_solver_t const&) = delete;

  index_t index_begin() const { return u_idx->index_begin(); }
  index_t index_end() const { return u_idx->index_end(); }
  derivative_storage_t uninit_derivative_storage() const { return {ctx}; }   // ddp.hpp:430-514 (allocated at create)

  multiplier_seq_t zero_multipliers() const {   // ddp.hpp:335-350: val = 0, jac = 0, origin = neutral configuration
    multiplier_seq_t m(eq_idx, nx, ndx);
    std::fill(m.m_origin.m_data.begin(), m.m_origin.m_data.end(), 0.0);
    std::fill(m.m_val_data.m_data.begin(), m.m_val_data.m_data.end(), 0.0);
    std::fill(m.m_jac_data.m_data.begin(), m.m_jac_data.m_data.end(), 0.0);
    return m;
  }

  // make_trajectory, ddp.hpp:392-415: x_0 = x_init, u_t = it_u(t), x_{t+1} = f(x_t, u_t)
  trajectory_t make_trajectory(std::function<void(index_t, scalar_t*)> it_u) const {
    trajectory_t traj(indexing::vec_regular_indexer(index_begin(), index_end() + 1, nx), u_idx);
    std::fill(traj.m_state_data.m_data.begin(), traj.m_state_data.m_data.end(), 0.0);
    for (index_t i = 0; i < nx; ++i) traj.x(0)[i] = x_init[static_cast<size_t>(i)];
    for (index_t t = index_begin(); t < index_end(); ++t) it_u(t, traj.u(t));
    upload_traj(traj, DDP_HIP_SEQ_X, DDP_HIP_SEQ_U);
    check(ddp_hip_rollout(ctx), "rollout");
    check(ddp_hip_download(ctx, DDP_HIP_SEQ_X, traj.m_state_data.data(), 0, 1), "download X");
    return traj;
  }

  void upload_traj(trajectory_t const& traj, int sx, int su) const {
    check(ddp_hip_upload(ctx, sx, traj.m_state_data.data(), 0, 1), "upload x");
    check(ddp_hip_upload(ctx, su, traj.m_control_data.data(), 0, 1), "upload u");
  }
  void upload_affine(affine_vector_function_seq_t<scalar_t> const& a, int so, int sv, int sj) const {
    check(ddp_hip_upload(ctx, so, a.m_origin.data(), 0, 1), "upload origin");
    if (a.m_val_data.size()) check(ddp_hip_upload(ctx, sv, a.m_val_data.data(), 0, 1), "upload val");
    if (a.m_jac_data.size()) check(ddp_hip_upload(ctx, sj, a.m_jac_data.data(), 0, 1), "upload jac");
  }

  // compute_derivatives (problem.hpp:956-998) along traj, into the resident storage
  void compute_derivatives(derivative_storage_t&, trajectory_t const& traj) const {
    upload_traj(traj, DDP_HIP_SEQ_X, DDP_HIP_SEQ_U);
    check(ddp_hip_linearize(ctx), "linearize");
  }

  // backward_pass<M>, ddp.hpp:845-853 / ddp_bwd.ipp:9-155
  template <method M>
  backward_pass_result_t<M> backward_pass(control_feedback_t&& ctrl_fb, trajectory_t const& current_traj, multiplier_seq_t const& mults,
                                          scalar_t regularization, scalar_t mu, derivative_storage_t const&) const {
    static_assert(M == method::primal_dual_affine_multipliers, "only the affine-multiplier method is instantiable (SURVEY.md App. C)");
    upload_traj(current_traj, DDP_HIP_SEQ_X, DDP_HIP_SEQ_U);
    upload_affine(mults, DDP_HIP_SEQ_MULT_ORIGIN, DDP_HIP_SEQ_MULT_VAL, DDP_HIP_SEQ_MULT_JAC);
    double reg = regularization, m = mu;
    check(ddp_hip_backward(ctx, &reg, &m, nullptr, 1000), "backward");
    check(ddp_hip_download(ctx, DDP_HIP_SEQ_FB_ORIGIN, ctrl_fb.m_origin.data(), 0, 1), "download");
    check(ddp_hip_download(ctx, DDP_HIP_SEQ_FB_VAL, ctrl_fb.m_val_data.data(), 0, 1), "download");
    check(ddp_hip_download(ctx, DDP_HIP_SEQ_FB_JAC, ctrl_fb.m_jac_data.data(), 0, 1), "download");
    return {std::move(ctrl_fb), m, reg};
  }

  // forward_pass<M>, ddp.hpp:855-862 / ddp_fwd.ipp:9-67
  template <method M>
  scalar_t forward_pass(trajectory_t& new_traj_storage, trajectory_t const& reference_traj, multiplier_seq_t const& old_mults,
                        backward_pass_result_t<M> const& bres, bool do_linesearch = true) const {
    upload_traj(reference_traj, DDP_HIP_SEQ_X, DDP_HIP_SEQ_U);
    upload_traj(new_traj_storage, DDP_HIP_SEQ_X_NEW, DDP_HIP_SEQ_U_NEW);
    upload_affine(old_mults, DDP_HIP_SEQ_MULT_ORIGIN, DDP_HIP_SEQ_MULT_VAL, DDP_HIP_SEQ_MULT_JAC);
    upload_affine(bres.feedback, DDP_HIP_SEQ_FB_ORIGIN, DDP_HIP_SEQ_FB_VAL, DDP_HIP_SEQ_FB_JAC);
    double mu = bres.mu, step = 0;
    // do_linesearch == false (ddp_fwd.ipp:61-63): n_alpha = 0, the full step taken unconditionally
    check(ddp_hip_forward(ctx, &mu, do_linesearch ? 8 : 0, &step, nullptr), "forward");
    check(ddp_hip_download(ctx, DDP_HIP_SEQ_X_NEW, new_traj_storage.m_state_data.data(), 0, 1), "download");
    check(ddp_hip_download(ctx, DDP_HIP_SEQ_U_NEW, new_traj_storage.m_control_data.data(), 0, 1), "download");
    return step;
  }

  // ---- outer loop (ddp.hpp:516-523, 576-627, 642-696, 745-842; mat_seq_common.hpp:62-89): the arithmetic runs on the
  // device (csrc/outer.hip) on the resident derivatives; only the small affine sequences travel -----------------------
  void download_affine(affine_vector_function_seq_t<scalar_t>& a, int so, int sv, int sj) const {
    check(ddp_hip_download(ctx, so, a.m_origin.data(), 0, 1), "download origin");
    if (a.m_val_data.size()) check(ddp_hip_download(ctx, sv, a.m_val_data.data(), 0, 1), "download val");
    if (a.m_jac_data.size()) check(ddp_hip_download(ctx, sj, a.m_jac_data.data(), 0, 1), "download jac");
  }
  struct optimality_t { scalar_t obj, constr; };
  optimality_t optimality(scalar_t mu) const {
    double m = mu, o = 0, c = 0;
    check(ddp_hip_optimality(ctx, &m, &o, &c), "optimality");
    return {o, c};
  }

  template <method M>
  std::pair<trajectory_t, control_feedback_t> solve(solver_parameters_t<scalar_t> sp, trajectory_t initial_trajectory,
                                                    std::vector<scalar_t> const* mult_jac_seed = nullptr, bool verbose = false) const {
    auto derivs = uninit_derivative_storage();
    auto& traj = initial_trajectory;
    auto new_traj = traj.clone();
    scalar_t reg = sp.reg, mu = sp.mu, w = sp.w, n = sp.n;
    auto mults = zero_multipliers();                                     // :759-764 (the reference draws jac at random; here an input)
    if (mult_jac_seed) mults.m_jac_data.m_data = *mult_jac_seed;
    for (index_t i = 0; i < T * nx; ++i) mults.m_origin.data()[i] = traj.m_state_data.data()[i];
    control_feedback_t ctrl_fb(u_idx, nx, ndx);
    compute_derivatives(derivs, traj);                                   // :768
    auto bres = backward_pass<M>(std::move(ctrl_fb), traj, mults, reg, mu, derivs);   // :769
    mu = bres.mu;                                                        // :771 (reg is not taken)
    scalar_t step = forward_pass<M>(new_traj, traj, mults, bres, true);  // :772
    ctrl_fb = std::move(bres.feedback);
    for (index_t iter = 0; iter < sp.max_iterations; ++iter) {
      compute_derivatives(derivs, traj);                                 // update_derivatives, :642-696
      upload_affine(mults, DDP_HIP_SEQ_MULT_ORIGIN, DDP_HIP_SEQ_MULT_VAL, DDP_HIP_SEQ_MULT_JAC);
      upload_affine(ctrl_fb, DDP_HIP_SEQ_FB_ORIGIN, DDP_HIP_SEQ_FB_VAL, DDP_HIP_SEQ_FB_JAC);
      check(ddp_hip_update_origin(ctx, 0), "update_origin(mults)");      // mat_seq_common.hpp:62-89
      check(ddp_hip_update_origin(ctx, 1), "update_origin(feedback)");
      auto opt = optimality(mu);                                         // :576-627, :516-523
      scalar_t opt_obj = opt.obj, opt_constr = opt.constr;
      if (verbose) std::printf("iter %3lld  opt obj %.3e  opt constr %.3e  mu %.3e  reg %.3e  step %.3e\n", (long long)iter, opt_obj, opt_constr, mu, reg, step);
      const bool done = opt_constr < sp.optimality_stopping_threshold && opt_obj < sp.optimality_stopping_threshold;
      if (!done && opt_obj < w) {
        if (opt_constr < n) {
          double m_ = mu;
          check(ddp_hip_update_multipliers(ctx, &m_), "update_multipliers");   // :680-688
          scalar_t oo = optimality(mu).obj;                               // :795-797
          n = oo / std::pow(mu, 0.1);
          w /= std::pow(mu, 1.0);
        } else {
          mu *= 10;                                                      // :791
        }
      }
      download_affine(mults, DDP_HIP_SEQ_MULT_ORIGIN, DDP_HIP_SEQ_MULT_VAL, DDP_HIP_SEQ_MULT_JAC);
      download_affine(ctrl_fb, DDP_HIP_SEQ_FB_ORIGIN, DDP_HIP_SEQ_FB_VAL, DDP_HIP_SEQ_FB_JAC);
      if (done) break;
      bres = backward_pass<M>(std::move(ctrl_fb), traj, mults, reg, mu, derivs);   // :804
      mu = bres.mu; reg = bres.reg;
      step = forward_pass<M>(new_traj, traj, mults, bres, true);         // :817
      ctrl_fb = std::move(bres.feedback);
      if (step >= 0.5) { reg /= 2; if (reg < 1e-5) reg = 0; }            // :819-824
      std::swap(traj, new_traj);                                         // :826
    }
    return {std::move(traj), std::move(ctrl_fb)};
  }
};

}  // namespace ddp
