// host/pendulum_ddp.cpp -- the reference's driver test/pendulum_ddp.cpp on the MI355X path: same problem (1-DoF
// pendulum m = l = 1, dt = 0.01, c = 1, target q = 3.14 at the unshifted time `horizon`, two time shifts), same call
// sequence (make_trajectory with zero controls, solve<primal_dual_affine_multipliers>), double instead of 1000-digit
// mpfr and double-appropriate solver parameters.  Prints the final state; exit code 0 when the
// terminal constraint is met to the penalty-method accuracy.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "ddp/ddp.hpp"

using namespace ddp;
using scalar_t = double;

int main(int argc, char** argv) {
  const index_t horizon = argc > 1 ? std::atoll(argv[1]) : 50;
  using model_t = pendulum_model_t<scalar_t>;
  auto model = model_t{1.0, 1.0};                                            // test/pendulum_ddp.cpp:30
  using dynamics_t = ddp::dynamics_t<model_t>;
  using problem_t = ddp::problem_t<dynamics_t>;

  constraint_t eq;                                                           // :35-50, :79-84
  eq.kind = DDP_HIP_EQ_CONFIG;
  eq.advance = 2;
  eq.m_eq_idx = indexing::shift_time_idx(indexing::shift_time_idx(
      indexing::range_row_filter(indexing::vec_regular_indexer(2, horizon + 2, 1), horizon, horizon + 1), 1), 1);
  eq.m_target = {3.14};

  std::vector<scalar_t> x_init = {0.0, 0.0};                                 // neutral configuration, zero velocity
  dynamics_t dy{model, 0.01};
  problem_t prob{0, horizon, 1.0, dy, eq};
  auto u_idx = indexing::vec_regular_indexer(0, horizon, 1);
  auto eq_idx = prob.m_constraint.eq_idx();

  try {
    ddp_solver_t<problem_t> solver{prob, u_idx, eq_idx, x_init};
    constexpr auto M = method::primal_dual_affine_multipliers;
    auto traj0 = solver.make_trajectory([](index_t, scalar_t* u) { u[0] = 0.0; });
    auto res = solver.solve<M>({40, 1e-8, 10.0, 0.0, 1.0, 10.0}, std::move(traj0), nullptr, true);
    auto& traj = res.first;
    std::printf("x_f: %.12f %.12f\n", traj.x_f()[0], traj.x_f()[1]);
    return std::fabs(traj.x_f()[0] - 3.14) < 0.05 ? 0 : 2;
  } catch (ddp_hip_error const& e) {
    std::printf("error: %s\n", e.what());
    return e.code == DDP_HIP_E_NODEVICE ? 77 : 1;
  }
}
