// ddp_hip_bridge.hpp -- glue between the reference's containers (s-elkazdadi/ddp-pinocchio, include/ddp/*) and the C-ABI
// of libddp_hip.so (include/ddp_hip/ddp_hip.h).  Used by the replacement ddp/ddp_bwd.ipp and ddp/ddp_fwd.ipp next to it.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE: it includes the reference's own headers, which need Eigen 3.3.7, Boost and fmt
// (absent here; the reference fetches them with conan over the network).  It is written against the reference as it stands
// -- every member it touches is cited -- and tests/test_adapters.py checks each ddp_hip_* call in these files against the
// declarations of ddp_hip.h (name and arity).  No stand-in headers are provided on purpose.
//
// To use: put this directory ahead of the reference's include/ on the include path (-I adapters -I <ref>/include), so that
// "ddp/ddp_bwd.ipp" and "ddp/ddp_fwd.ipp" (included by test/pinocchio_ddp.cpp:6-7) resolve here, add -I <repo>/include,
// link libddp_hip.so, and build adapters/pinocchio_double.cpp instead of the reference's empty test/pinocchio_double.cpp.
#ifndef DDP_HIP_BRIDGE_HPP
#define DDP_HIP_BRIDGE_HPP

#include "ddp/ddp.hpp"          // the reference's: ddp_solver_t (ddp.hpp:300-869), trajectory_t (trajectory.hpp:9-113)
#include "ddp_hip/ddp_hip.h"

#include <map>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace ddp {
namespace hip_bridge {

inline void check(int rc, char const* where) {
  // the reference reports failures through DDP_ASSERT (-> std::terminate, src/lib.cpp:107-110); the library returns codes
  if (rc < 0) throw std::runtime_error(std::string(where) + ": " + ddp_hip_strerror(rc));
}

// The model table of the problem's dynamics.  model_t<double> (adapters/pinocchio_double.cpp) registers its tables by
// address -- the reference's header declares no accessor and the drop-in does not touch it; pendulum_model_t
// (pendulum_model.hpp:10-133) is recognised by its two scalars.
void model_table_of(void const* model, ddp_hip_model& out, ddp_hip_model_storage& st);   // adapters/pinocchio_double.cpp
void frame_of(void const* model, index_t frame_id, int32_t& joint, double off[3]);
template <typename Model>
void fill_model(Model const& model, ddp_hip_model& out, ddp_hip_model_storage& st) {
  model_table_of(static_cast<void const*>(&model), out, st);
}
template <typename Scalar>
void fill_model(pendulum_model_t<Scalar> const& model, ddp_hip_model& out, ddp_hip_model_storage& st) {
  check(ddp_hip_builtin_model(DDP_HIP_BUILTIN_PENDULUM, 0, &st, &out), "ddp_hip_builtin_model");
  // m_mass / m_length are PRIVATE (pendulum_model.hpp:20-26) and the drop-in does not touch the reference's headers: both
  // are read back through the public dynamics_aba (pendulum_model.hpp:105-115),
  //   acc(q, v, tau) = -g / length * sin(q) + tau / mass,   g = 9.81 (:26)
  // probed at (q, tau) = (0, 1) -> 1 / mass and at (pi / 2, 0) -> -g / length
  using mat_t = Eigen::Matrix<Scalar, 1, 1>;
  mat_t acc, q, v, tau;
  v[0] = 0;
  q[0] = 0; tau[0] = 1;
  model.dynamics_aba(eigen::as_mut_view(acc), eigen::as_const_view(q), eigen::as_const_view(v), eigen::as_const_view(tau));
  out.mass = 1.0 / static_cast<double>(acc[0]);
  q[0] = static_cast<Scalar>(1.5707963267948966192313216916398L); tau[0] = 0;
  model.dynamics_aba(eigen::as_mut_view(acc), eigen::as_const_view(q), eigen::as_const_view(v), eigen::as_const_view(tau));
  out.length = -9.81 / static_cast<double>(acc[0]);
}

// constraint description.  The reference composes constraint types (problem.hpp:527-870); the adapter needs to know which
// base constraint sits under how many constraint_advance_time_t wrappers.  Specialise for other compositions.
template <typename C>
struct constraint_traits;   // kind, advance, target(c, t, out), frame_joint(c), frame_off(c)

template <typename C>
struct constraint_traits<constraint_advance_time_t<C>> {      // problem.hpp:527-624
  static constexpr int kind = constraint_traits<C>::kind;
  static constexpr int advance = constraint_traits<C>::advance + 1;
  static auto base(constraint_advance_time_t<C> const& c) -> decltype(constraint_traits<C>::base(c.m_constraint)) {
    return constraint_traits<C>::base(c.m_constraint);
  }
};
template <typename Model, typename View>
struct constraint_traits<config_constraint_t<Model, View>> {  // problem.hpp:744-864
  static constexpr int kind = DDP_HIP_EQ_CONFIG;
  static constexpr int advance = 0;
  static auto base(config_constraint_t<Model, View> const& c) -> config_constraint_t<Model, View> const& { return c; }
};
template <typename Model, typename View>
struct constraint_traits<spatial_constraint_t<Model, View>> { // problem.hpp:631-742
  static constexpr int kind = DDP_HIP_EQ_FRAME;
  static constexpr int advance = 0;
  static auto base(spatial_constraint_t<Model, View> const& c) -> spatial_constraint_t<Model, View> const& { return c; }
};

template <typename Model, typename View>
void set_frame(ddp_hip_problem& hp, spatial_constraint_t<Model, View> const& c) {
  // the library attaches the frame to a joint + an offset in that joint's frame; hip_bridge::frame_of (adapters/
  // pinocchio_double.cpp) resolves the reference's frame index (problem.hpp:741 m_frame_id)
  frame_of(static_cast<void const*>(&c.m_dynamics.m_model), c.m_frame_id, hp.frame_joint, hp.frame_off);
}
template <typename C>
void set_frame(ddp_hip_problem&, C const&) {}

struct entry_t {
  ddp_hip_ctx* ctx = nullptr;
  bool derivatives_resident = false;   // set by the B3 seam (compute_derivatives on the device): backward_pass uploads no derivative
};

inline auto registry() -> std::map<void const*, entry_t>& {
  static std::map<void const*, entry_t> r;
  return r;
}
inline auto registry_mutex() -> std::mutex& {
  static std::mutex m;
  return m;
}

// One context per solver object (ddp_solver_t is neither copyable nor movable, ddp.hpp:737-741, so its address is a key).
template <typename Solver>
auto entry_for(Solver const& solver) -> entry_t& {
  using scalar_t = typename Solver::scalar_t;
  static_assert(std::is_same<scalar_t, double>::value, "the HIP path is double only (the reference's drivers default to mpfr: flip their #if, test/pinocchio_ddp.cpp:14-20)");
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto& e = registry()[static_cast<void const*>(&solver)];
  if (e.ctx) return e;

  auto const& prob = solver.prob;                               // ddp.hpp:864
  index_t const begin = solver.index_begin(), end = solver.index_end();
  if (begin != 0) throw std::invalid_argument("index_begin must be 0 (ddp_bwd.ipp:149)");
  ddp_hip_problem hp{};
  ddp_hip_model_storage st;
  fill_model(prob.m_dynamics.m_model, hp.model, st);            // problem.hpp:520
  hp.dt = static_cast<double>(prob.m_dynamics.dt);              // problem.hpp:522
  hp.c = static_cast<double>(prob.c);                           // problem.hpp:1147
  hp.T = end - begin;
  hp.batch = 1;
  using traits = constraint_traits<typename Solver::problem_t::constraint_t>;
  hp.eq_kind = traits::kind;
  hp.eq_advance = traits::advance;
  std::vector<std::int64_t> ne(static_cast<std::size_t>(hp.T));
  std::vector<double> target;
  auto const& base = traits::base(prob.m_constraint);
  for (index_t t = begin; t < end; ++t) {
    index_t const e_t = solver.eq_idx.rows(t).value();          // rows at SOLVER time t (the indexer is already time-shifted, problem.hpp:553)
    ne[static_cast<std::size_t>(t)] = e_t;
    auto tg = eigen::as_const_view(base.m_constraint_target_view[t + traits::advance]);   // problem.hpp:680,793: indexed by unshifted time
    for (index_t i = 0; i < e_t; ++i) target.push_back(static_cast<double>(tg[i]));
  }
  hp.ne = ne.data();
  hp.eq_target = target.data();
  set_frame(hp, base);
  // dynamics_t::second_order_finite_diff (problem.hpp:523): true -> mode 2 (:152-298), false -> mode 1 (:67-150).
  // first_order_deriv is analytic in the reference (problem.hpp:463-503): first_order_fd = 0.  The library takes this
  // combination for every vector-space model, constrained or not (lin_analytic.hip: ana_eq_kernel runs the chain rule of
  // problem.hpp:569-605 and its mode-1 differences :611-620 on the analytic jacobians; K <= 2 time shifts on large trees)
  hp.first_order_fd = 0;
  hp.fd_mode = prob.m_dynamics.second_order_finite_diff ? 2 : 1;
  check(ddp_hip_create(&hp, /*device=*/0, /*flags=*/0, &e.ctx), "ddp_hip_create");
  return e;
}
template <typename Solver>
auto context_for(Solver const& solver) -> ddp_hip_ctx* { return entry_for(solver).ctx; }

inline void release(void const* solver) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto it = registry().find(solver);
  if (it == registry().end()) return;
  if (it->second.ctx) ddp_hip_destroy(it->second.ctx);
  registry().erase(it);
}

// every mat_seq_t / tensor_seq_t owns ONE contiguous Eigen::VectorX (detail/mat_seq.hpp:22-23, detail/tensor.hpp:360-361) in
// exactly the flat layout the library keeps resident: the whole sequence moves with one call
template <typename Seq>
void upload(ddp_hip_ctx* ctx, int seq, Seq const& s) {
  if (s.m_data.size() == 0) return;
  check(ddp_hip_upload(ctx, seq, s.m_data.data(), 0, 1), "ddp_hip_upload");
}
template <typename Seq>
void download(ddp_hip_ctx* ctx, int seq, Seq& s) {
  if (s.m_data.size() == 0) return;
  check(ddp_hip_download(ctx, seq, s.m_data.data(), 0, 1), "ddp_hip_download");
}
inline void upload_raw(ddp_hip_ctx* ctx, int seq, double const* p) { check(ddp_hip_upload(ctx, seq, p, 0, 1), "ddp_hip_upload"); }

// affine_vector_function_seq_t: m_origin, m_val_data, m_jac_data (detail/mat_seq_common.hpp:25-27)
template <typename Affine>
void upload_affine(ddp_hip_ctx* ctx, int seq_origin, Affine const& a) {
  upload(ctx, seq_origin, a.m_origin);
  upload(ctx, seq_origin + 1, a.m_val_data);
  upload(ctx, seq_origin + 2, a.m_jac_data);
}
template <typename Affine>
void download_affine(ddp_hip_ctx* ctx, int seq_origin, Affine& a) {
  download(ctx, seq_origin, a.m_origin);
  download(ctx, seq_origin + 1, a.m_val_data);
  download(ctx, seq_origin + 2, a.m_jac_data);
}
static_assert(DDP_HIP_SEQ_MULT_VAL == DDP_HIP_SEQ_MULT_ORIGIN + 1 && DDP_HIP_SEQ_MULT_JAC == DDP_HIP_SEQ_MULT_ORIGIN + 2, "");
static_assert(DDP_HIP_SEQ_FB_VAL == DDP_HIP_SEQ_FB_ORIGIN + 1 && DDP_HIP_SEQ_FB_JAC == DDP_HIP_SEQ_FB_ORIGIN + 2, "");

// trajectory_t: m_state_data, m_control_data (trajectory.hpp:22-23)
template <typename Traj>
void upload_traj(ddp_hip_ctx* ctx, int seq_x, int seq_u, Traj const& traj) {
  upload(ctx, seq_x, traj.m_state_data);
  upload(ctx, seq_u, traj.m_control_data);
}

// derivative_storage_t (ddp.hpp:52-245): lfx / lfxx are plain Eigen matrices, the rest are sequences
template <typename Derivs>
void upload_derivatives(ddp_hip_ctx* ctx, Derivs const& d) {
  upload_raw(ctx, DDP_HIP_SEQ_LFX, d.lfx.data());
  upload_raw(ctx, DDP_HIP_SEQ_LFXX, d.lfxx.data());
  upload(ctx, DDP_HIP_SEQ_LX, d.lx);   upload(ctx, DDP_HIP_SEQ_LU, d.lu);
  upload(ctx, DDP_HIP_SEQ_LXX, d.lxx); upload(ctx, DDP_HIP_SEQ_LUX, d.lux); upload(ctx, DDP_HIP_SEQ_LUU, d.luu);
  upload(ctx, DDP_HIP_SEQ_F_VAL, d.f_val);
  upload(ctx, DDP_HIP_SEQ_FX, d.fx);   upload(ctx, DDP_HIP_SEQ_FU, d.fu);
  upload(ctx, DDP_HIP_SEQ_FXX, d.fxx); upload(ctx, DDP_HIP_SEQ_FUX, d.fux); upload(ctx, DDP_HIP_SEQ_FUU, d.fuu);
  upload(ctx, DDP_HIP_SEQ_EQ_VAL, d.eq_val);
  upload(ctx, DDP_HIP_SEQ_EQ_X, d.eq_x);   upload(ctx, DDP_HIP_SEQ_EQ_U, d.eq_u);
  upload(ctx, DDP_HIP_SEQ_EQ_XX, d.eq_xx); upload(ctx, DDP_HIP_SEQ_EQ_UX, d.eq_ux); upload(ctx, DDP_HIP_SEQ_EQ_UU, d.eq_uu);
}

// Seam B3: problem_t::compute_derivatives (problem.hpp:956-998) on the device.  Call this instead of
// prob.compute_derivatives(derivs, traj) at ddp.hpp:655,768: the derivative sequences then stay resident and
// backward_pass uploads only the multipliers.
template <typename Solver, typename Traj>
void compute_derivatives_resident(Solver const& solver, Traj const& traj) {
  auto& e = entry_for(solver);
  upload_traj(e.ctx, DDP_HIP_SEQ_X, DDP_HIP_SEQ_U, traj);
  check(ddp_hip_linearize(e.ctx), "ddp_hip_linearize");
  e.derivatives_resident = true;
}

}  // namespace hip_bridge
}  // namespace ddp
#endif
