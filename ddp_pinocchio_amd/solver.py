"""solve<primal_dual_affine_multipliers> (reference include/ddp/ddp.hpp:745-842) for every instance of a context,
driven from Python over the C-ABI.  Everything that touches a sequence runs on the device (linearise, backward and
forward sweeps, update_origin, optimality measures, multiplier update); the host keeps only the per-instance scalars
(mu, reg, w, n, step) and applies the reference's scalar rules to them.  The C++ mirror of the same loop is
include/ddp/ddp.hpp (ddp_solver_t::solve)."""
import numpy as np


def solve(ctx, max_iterations, threshold, mu, reg, w, n, n_alpha=8, max_restarts=1000):
    """X / U hold the initial trajectory, X_NEW / U_NEW a clone of it (ddp.hpp:752), MULT_* the initial multipliers
    (val = 0, jac = the seed the reference draws at random, origin = X: ddp.hpp:759-764).  Returns a log dict of
    per-instance arrays; the final trajectory is left in X / U, the feedback in FB_*."""
    B = ctx.batch
    mu = np.full(B, float(mu)); reg = np.full(B, float(reg)); w = np.full(B, float(w)); n = np.full(B, float(n))
    ctx.linearize()                                                    # :768
    _, _reg_b, mu, _ = ctx.backward(reg, mu, max_restarts)             # :769-771 (mu is taken, reg is not)
    _, step, _ = ctx.forward(mu, n_alpha=n_alpha)                      # :772
    done = np.zeros(B, dtype=bool)
    it = 0
    opt_obj = np.zeros(B); opt_constr = np.zeros(B)
    for it in range(max_iterations):
        ctx.linearize()                                                # update_derivatives, :642-696
        ctx.update_origin(0)
        ctx.update_origin(1)
        opt_obj, opt_constr = ctx.optimality(mu)
        done = (opt_constr < threshold) & (opt_obj < threshold)        # :673-675
        if done.all():
            break
        upd = ~done & (opt_obj < w) & (opt_constr < n)
        grow = ~done & (opt_obj < w) & ~(opt_constr < n)
        if upd.any():
            # instances that do not update keep their multipliers: a zero step size for them
            ctx.update_multipliers(np.where(upd, mu, 0.0))             # :680-688
            oo, _ = ctx.optimality(mu)                                 # :795-797
            n = np.where(upd, oo / mu ** 0.1, n)
            w = np.where(upd, w / mu, w)
        mu = np.where(grow, mu * 10, mu)                               # :791
        _, reg, mu, _ = ctx.backward(reg, mu, max_restarts)            # :804-806
        _, step, _ = ctx.forward(mu, n_alpha=n_alpha)                  # :817
        reg = np.where(step >= 0.5, reg / 2, reg)                      # :819-824
        reg = np.where(reg < 1e-5, 0.0, reg)
        ctx.swap_traj()                                                # :826
    return {"iterations": it, "done": done, "mu": mu, "reg": reg, "w": w, "n": n, "last_step": step,
            "opt_obj": opt_obj, "opt_constr": opt_constr}
