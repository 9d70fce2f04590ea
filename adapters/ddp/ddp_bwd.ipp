// ddp/ddp_bwd.ipp -- drop-in replacement of the reference's include/ddp/ddp_bwd.ipp (:9-155): the out-of-line definition
// of ddp_solver_t<Problem>::backward_pass<M> declared at include/ddp/ddp.hpp:845-853, forwarding to libddp_hip.so.
// Not compiled in this repository's image (needs the reference's Eigen / Boost / fmt): see ddp_hip_bridge.hpp.
#ifndef DDP_IMPL_HPP_UBVAKU5V
#define DDP_IMPL_HPP_UBVAKU5V   // the reference's own include guard: whichever file comes first on the include path wins

#include "ddp/ddp.hpp"
#include "ddp_hip_bridge.hpp"

namespace ddp {

template <typename Problem>
template <method M>
auto ddp_solver_t<Problem>::
    // clang-format off
  backward_pass(
      control_feedback_t&&                            ctrl_fb,
      trajectory_t const&                             current_traj,
      typename multiplier_sequence<M>::type const&    mults,
      scalar_t                                        regularization,
      scalar_t                                        mu,
      derivative_storage_t const&                     derivatives
  ) const -> backward_pass_result_t<M>
{
  // clang-format on
  static_assert(M == method::primal_dual_affine_multipliers, "the only instantiable method (SURVEY.md Appendix C)");
  auto& e = hip_bridge::entry_for(*this);
  ddp_hip_ctx* ctx = e.ctx;

  // fb.origin(t) = x_t is read from the resident trajectory (ddp_bwd.ipp:134)
  hip_bridge::upload_traj(ctx, DDP_HIP_SEQ_X, DDP_HIP_SEQ_U, current_traj);
  if (not e.derivatives_resident) hip_bridge::upload_derivatives(ctx, derivatives);   // 1.2 GB of tensors at the Talos shape: prefer seam B3
  hip_bridge::upload_affine(ctx, DDP_HIP_SEQ_MULT_ORIGIN, mults.eq);

  double reg = static_cast<double>(regularization), m = static_cast<double>(mu);
  // the reference's while(!success) is unbounded (ddp_bwd.ipp:26); 1000 doublings of mu overflow a double anyway
  int rc = ddp_hip_backward(ctx, &reg, &m, nullptr, /*max_restarts=*/1000);
  hip_bridge::check(rc, "ddp_hip_backward");                                          // rc > 0: the sweep restarted (:105-132)

  hip_bridge::download_affine(ctx, DDP_HIP_SEQ_FB_ORIGIN, ctrl_fb);
  return {DDP_MOVE(ctrl_fb), scalar_t(m), scalar_t(reg)};                             // ddp_bwd.ipp:154
}

}  // namespace ddp
#endif
