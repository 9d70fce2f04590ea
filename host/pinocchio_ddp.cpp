// host/pinocchio_ddp.cpp -- the reference's driver test/pinocchio_ddp.cpp on the MI355X path: a 6-revolute UR5-like arm
// (the built-in table stands in for ur5_gripper.urdf: no URDF and no urdfdom offline), horizon 10 (:33), dt = 0.01,
// c = 1, configuration constraint to the neutral q at every step through two time shifts (:35-49, :60-68), zero initial
// controls (:72-84), solve<primal_dual_affine_multipliers> (:99).  Differences: double instead of 1000-digit mpfr with
// double-appropriate solver parameters; forward-difference first order + mode-2 second order instead of
// Pinocchio's analytic ABA derivatives + mode 1.  The arm starts away from the neutral configuration so
// that the constraint has work to do.  With forward-difference jacobians the stationarity measure bottoms out near
// 1e-5, so the reference's multiplier schedule (w /= mu per update) stops updating after two rounds: the loop ends as a
// penalty method.  Exit code 0 when the terminal constraint violation was at least halved and nothing failed.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "ddp/ddp.hpp"

using namespace ddp;
using scalar_t = double;

int main(int argc, char** argv) {
  const index_t horizon = argc > 1 ? std::atoll(argv[1]) : 10;
  using model_t = pinocchio::model_t<scalar_t>;
  auto model = model_t{DDP_HIP_BUILTIN_CHAIN6};
  const index_t nq = model.configuration_dim(), nv = model.tangent_dim();
  using dynamics_t = ddp::dynamics_t<model_t>;
  using problem_t = ddp::problem_t<dynamics_t>;

  constraint_t eq;
  eq.kind = DDP_HIP_EQ_CONFIG;
  eq.advance = 2;
  eq.m_eq_idx = indexing::shift_time_idx(indexing::shift_time_idx(indexing::vec_regular_indexer(2, horizon + 2, nq), 1), 1);
  eq.m_target.assign(static_cast<size_t>(nq * horizon), 0.0);                 // neutral configuration

  std::vector<scalar_t> x_init(static_cast<size_t>(nq + nv), 0.0);
  for (index_t i = 0; i < nq; ++i) x_init[static_cast<size_t>(i)] = 0.05 * (i % 2 ? 1 : -1);
  dynamics_t dy{model, 0.01, true};
  problem_t prob{0, horizon, 1.0, dy, eq};
  auto u_idx = indexing::vec_regular_indexer(0, horizon, nv);
  auto eq_idx = prob.m_constraint.eq_idx();

  try {
    ddp_solver_t<problem_t> solver{prob, u_idx, eq_idx, x_init};
    constexpr auto M = method::primal_dual_affine_multipliers;
    auto traj0 = solver.make_trajectory([nv](index_t, scalar_t* u) { for (index_t i = 0; i < nv; ++i) u[i] = 0.0; });
    auto viol = [&](ddp_solver_t<problem_t>::trajectory_t& tr) {
      scalar_t s = 0;
      for (index_t i = 0; i < nq; ++i) s = std::max(s, std::fabs(tr.x_f()[i]));
      return s;
    };
    const scalar_t v0 = viol(traj0);
    const scalar_t mu0 = argc > 2 ? std::atof(argv[2]) : 1e3, w0 = argc > 3 ? std::atof(argv[3]) : 1e-1, n0 = argc > 4 ? std::atof(argv[4]) : 10.0;
    const index_t iters = argc > 5 ? std::atoll(argv[5]) : 20;
    auto res = solver.solve<M>({iters, 1e-8, mu0, 0.0, w0, n0}, std::move(traj0), nullptr, true);
    const scalar_t v1 = viol(res.first);
    std::printf("max |q_T - q_neutral|: %.3e -> %.3e\n", v0, v1);
    return v1 < 0.5 * v0 ? 0 : 2;
  } catch (ddp_hip_error const& e) {
    std::printf("error: %s\n", e.what());
    return e.code == DDP_HIP_E_NODEVICE ? 77 : 1;
  }
}
