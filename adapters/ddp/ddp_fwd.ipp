// ddp/ddp_fwd.ipp -- drop-in replacement of the reference's include/ddp/ddp_fwd.ipp (:9-67): the out-of-line definition of
// ddp_solver_t<Problem>::forward_pass<M> declared at include/ddp/ddp.hpp:855-862, forwarding to libddp_hip.so.
// Not compiled in this repository's image (needs the reference's Eigen / Boost / fmt): see ddp_hip_bridge.hpp.
#ifndef DDP_FWD_TCC_O5KLTLOB
#define DDP_FWD_TCC_O5KLTLOB    // the reference's own include guard

#include "ddp/ddp.hpp"
#include "ddp_hip_bridge.hpp"

namespace ddp {

// clang-format off
template <typename Problem>
template <method M>
auto ddp_solver_t<Problem>::
  forward_pass(
      trajectory_t&                                   new_traj_storage,
      trajectory_t const&                             reference_traj,
      typename multiplier_sequence<M>::type const&    old_mults,
      backward_pass_result_t<M> const&                backward_pass_result,
      bool                                            do_linesearch
  ) const -> scalar_t {
  // clang-format on
  ddp_hip_ctx* ctx = hip_bridge::context_for(*this);
  hip_bridge::upload_traj(ctx, DDP_HIP_SEQ_X, DDP_HIP_SEQ_U, reference_traj);
  // x_new,0 is never written by the pass: the caller presets it (new_traj = traj.clone(), ddp.hpp:752)
  hip_bridge::upload_traj(ctx, DDP_HIP_SEQ_X_NEW, DDP_HIP_SEQ_U_NEW, new_traj_storage);
  hip_bridge::upload_affine(ctx, DDP_HIP_SEQ_MULT_ORIGIN, old_mults.eq);
  hip_bridge::upload_affine(ctx, DDP_HIP_SEQ_FB_ORIGIN, backward_pass_result.feedback);

  double mu = static_cast<double>(backward_pass_result.mu), step = 0;
  // 8 halvings per round, the largest accepted one kept: the decision of the sequential halving (ddp_fwd.ipp:29-64).
  // Without a line search (do_linesearch == false, :61-63) the reference takes the full step unconditionally:
  // n_alpha = 0 asks the library for exactly that (one rollout at step 1, accepted whatever the cost does)
  int rc = ddp_hip_forward(ctx, &mu, do_linesearch ? 8 : 0, &step, nullptr);
  hip_bridge::check(rc, "ddp_hip_forward");                       // rc > 0: step < 1e-10 was reached (:35-37)

  hip_bridge::download(ctx, DDP_HIP_SEQ_X_NEW, new_traj_storage.m_state_data);
  hip_bridge::download(ctx, DDP_HIP_SEQ_U_NEW, new_traj_storage.m_control_data);
  return scalar_t(step);
}

}  // namespace ddp
#endif
