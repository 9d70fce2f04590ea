// solve.cpp -- solve<primal_dual_affine_multipliers> (reference include/ddp/ddp.hpp:745-842) for every instance of a
// context, as one C-ABI call.  Every sequence operation runs on the device through the library's own entry points
// (linearise, backward / forward sweeps, update_origin, optimality measures, multiplier update); this file holds only the
// per-instance scalars (mu, reg, w, n, step) and the reference's scalar rules, applied instance by instance.
//
// Per-problem semantics in a batch: the reference returns a problem at its first optimum (ddp.hpp:799-800).  Here such
// an instance is latched and frozen (ddp_hip_set_active): the sweeps skip it and swap_traj keeps its trajectory, so its
// result and its iteration count do not depend on its batch-mates.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "ddp_hip/ddp_hip.h"

// ddp_hip_ctx is opaque here: only the public C-ABI is used
extern "C" int ddp_hip_solve(ddp_hip_ctx* ctx, const ddp_hip_solver_params* sp, ddp_hip_solve_log* log) {
  if (!ctx || !sp || !log || sp->max_iterations < 0 || sp->n_alpha < 1) return DDP_HIP_E_ARG;
  const int64_t B = ddp_hip_batch(ctx);
  if (B < 1) return DDP_HIP_E_ARG;
  const size_t nb = (size_t)B;
  std::vector<double> mu(nb, sp->mu), reg(nb, sp->reg), w(nb, sp->w), n(nb, sp->n), step(nb, 0.0);
  std::vector<double> obj(nb, 0.0), constr(nb, 0.0), tmp_reg(nb), tmp(nb), oo(nb), cc(nb);
  std::vector<int32_t> active(nb, 1);
  std::vector<int64_t> iters(nb, sp->max_iterations);
  std::vector<int32_t> result(nb, 0);
  int rc, ev = DDP_HIP_OK;
#define SOLVE_TRY(expr)                                          \
  do {                                                           \
    rc = (expr);                                                 \
    if (rc < 0) { (void)ddp_hip_set_active(ctx, nullptr); (void)ddp_hip_set_async(ctx, 0); return rc; } \
    if (rc > 0) ev |= rc;                                        \
  } while (0)

  SOLVE_TRY(ddp_hip_set_active(ctx, nullptr));
  // the loop enqueues: only the calls that return values to the rules below wait for the device (ddp_hip_set_async)
  SOLVE_TRY(ddp_hip_set_async(ctx, getenv("DDP_HIP_SOLVE_SYNC") ? 0 : 1));   // (DDP_HIP_SOLVE_SYNC: development A/B, every call waits as in round 2)
  SOLVE_TRY(ddp_hip_linearize(ctx));                                                        // :768
  tmp_reg = reg;
  SOLVE_TRY(ddp_hip_backward(ctx, tmp_reg.data(), mu.data(), nullptr, sp->max_restarts));   // :769-771 (mu is taken, reg is not)
  SOLVE_TRY(ddp_hip_forward(ctx, mu.data(), sp->n_alpha, step.data(), nullptr));            // :772
  for (int64_t it = 0; it < sp->max_iterations; ++it) {
    SOLVE_TRY(ddp_hip_linearize(ctx));                                                      // update_derivatives, :642-696
    SOLVE_TRY(ddp_hip_update_origin(ctx, 0));                                               // :657
    SOLVE_TRY(ddp_hip_update_origin(ctx, 1));                                               // :658
    SOLVE_TRY(ddp_hip_optimality(ctx, mu.data(), oo.data(), cc.data()));                    // :660-661
    bool any_active = false, any_upd = false;
    for (size_t b = 0; b < nb; ++b) {
      if (!active[b]) continue;
      obj[b] = oo[b]; constr[b] = cc[b];
      if (cc[b] < sp->optimality_stopping_threshold && oo[b] < sp->optimality_stopping_threshold) {   // :673-675
        active[b] = 0; result[b] = 1; iters[b] = it;                                        // :799-800: returned as it is now
      } else any_active = true;
    }
    SOLVE_TRY(ddp_hip_set_active(ctx, active.data()));
    if (!any_active) break;
    // multiplier update attempt (:677-695) and its consequences (:786-798)
    for (size_t b = 0; b < nb; ++b) {
      const bool upd = active[b] && obj[b] < w[b] && constr[b] < n[b];
      tmp[b] = upd ? mu[b] : 0.0;            // instances that do not update keep their multipliers: a zero step for them
      any_upd |= upd;
    }
    if (any_upd) {
      SOLVE_TRY(ddp_hip_update_multipliers(ctx, tmp.data()));                               // :680-688
      SOLVE_TRY(ddp_hip_optimality(ctx, mu.data(), oo.data(), cc.data()));                  // :795
    }
    for (size_t b = 0; b < nb; ++b) {
      if (!active[b] || !(obj[b] < w[b])) continue;                                         // no_update
      if (constr[b] < n[b]) { n[b] = oo[b] / pow(mu[b], 0.1); w[b] /= pow(mu[b], 1.0); }    // update_success, :796-797
      else mu[b] *= 10;                                                                     // update_failure, :791
    }
    SOLVE_TRY(ddp_hip_backward(ctx, reg.data(), mu.data(), nullptr, sp->max_restarts));     // :804-806
    SOLVE_TRY(ddp_hip_forward(ctx, mu.data(), sp->n_alpha, step.data(), nullptr));          // :817
    for (size_t b = 0; b < nb; ++b)
      if (active[b] && step[b] >= 0.5) { reg[b] /= 2; if (reg[b] < 1e-5) reg[b] = 0; }      // :819-824
    SOLVE_TRY(ddp_hip_swap_traj(ctx));                                                      // :826
  }
#undef SOLVE_TRY
  (void)ddp_hip_set_active(ctx, nullptr);
  { const int rc_ = ddp_hip_set_async(ctx, 0); if (rc_ < 0) return rc_; }                  // (waits for the stream)
  for (size_t b = 0; b < nb; ++b) {
    log[b].iterations = iters[b]; log[b].result = result[b]; log[b].pad_ = 0;
    log[b].mu = mu[b]; log[b].reg = reg[b]; log[b].w = w[b]; log[b].n = n[b];
    log[b].last_step = step[b]; log[b].opt_obj = obj[b]; log[b].opt_constr = constr[b];
  }
  return ev;
}
