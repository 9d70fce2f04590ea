"""solve<primal_dual_affine_multipliers> (reference include/ddp/ddp.hpp:745-842) for every instance of a context.

The loop itself lives in the library (`ddp_hip_solve`, csrc/solve.cpp: every sequence operation on the device, the
per-instance scalars mu / reg / w / n / step and the reference's scalar rules on the host side of the C-ABI, instances
latched and frozen at their first optimum like the reference's `return` at ddp.hpp:799-800).  `solve` below is the thin
binding; `solve_stepwise` drives the same entry points one by one from Python and exists so that the tests can hold the
two against each other.  The C++ mirror for a single problem is include/ddp/ddp.hpp (ddp_solver_t::solve)."""
import numpy as np


def solve(ctx, max_iterations, threshold, mu, reg, w, n, n_alpha=8, max_restarts=1000):
    """X / U hold the initial trajectory, X_NEW / U_NEW a clone of it (ddp.hpp:752), MULT_* the initial multipliers
    (val = 0, jac = the seed the reference draws at random, origin = X: ddp.hpp:759-764).  Returns a log dict of
    per-instance arrays (iterations, done, mu, reg, w, n, last_step, opt_obj, opt_constr); the final trajectory is left
    in X / U, the feedback in FB_*."""
    _, log = ctx.solve(max_iterations, threshold, mu, reg, w, n, n_alpha=n_alpha, max_restarts=max_restarts)
    return log


def solve_stepwise(ctx, max_iterations, threshold, mu, reg, w, n, n_alpha=8, max_restarts=1000):
    """The same loop, one C-ABI call per operation (cross-check of ddp_hip_solve)."""
    B = ctx.batch
    mu = np.full(B, float(mu)); reg = np.full(B, float(reg)); w = np.full(B, float(w)); n = np.full(B, float(n))
    ctx.set_active(None)
    ctx.linearize()                                                    # :768
    _, _reg_b, mu, _ = ctx.backward(reg, mu, max_restarts)             # :769-771 (mu is taken, reg is not)
    _, step, _ = ctx.forward(mu, n_alpha=n_alpha)                      # :772
    active = np.ones(B, dtype=bool)
    iters = np.full(B, max_iterations)
    opt_obj = np.zeros(B); opt_constr = np.zeros(B)
    for it in range(max_iterations):
        ctx.linearize()                                                # update_derivatives, :642-696
        ctx.update_origin(0)
        ctx.update_origin(1)
        oo, cc = ctx.optimality(mu)
        opt_obj = np.where(active, oo, opt_obj); opt_constr = np.where(active, cc, opt_constr)
        now_done = active & (cc < threshold) & (oo < threshold)        # :673-675, returned as it is now (:799-800)
        iters = np.where(now_done, it, iters)
        active &= ~now_done
        ctx.set_active(active.astype(np.int32))
        if not active.any():
            break
        upd = active & (opt_obj < w) & (opt_constr < n)
        grow = active & (opt_obj < w) & ~(opt_constr < n)
        if upd.any():
            ctx.update_multipliers(np.where(upd, mu, 0.0))             # :680-688 (a zero step for the others)
            oo, _ = ctx.optimality(mu)                                 # :795-797
            n = np.where(upd, oo / mu ** 0.1, n)
            w = np.where(upd, w / mu, w)
        mu = np.where(grow, mu * 10, mu)                               # :791
        _, reg, mu, _ = ctx.backward(reg, mu, max_restarts)            # :804-806
        _, step_new, _ = ctx.forward(mu, n_alpha=n_alpha)              # :817
        step = np.where(active, step_new, step)
        half = np.where(reg / 2 < 1e-5, 0.0, reg / 2)                  # :819-824: the floor applies inside the step >= 0.5 branch
        reg = np.where(active & (step >= 0.5), half, reg)
        ctx.swap_traj()                                                # :826
    ctx.set_active(None)
    return {"iterations": iters, "done": ~active, "mu": mu, "reg": reg, "w": w, "n": n, "last_step": step,
            "opt_obj": opt_obj, "opt_constr": opt_constr}
