"""CPU restatement of the projected-gradient-descent loop shared by the reference's refactored
PDECO drivers.

TEST INFRASTRUCTURE (see oracle/__init__.py).

  nonlinear_FCT_PDECO_refactored.py:105-232        (one state, final-time misfit)
  Schnak_FCT_PDECO_refactored.py:122-259           (two states, final-time misfit)
  chemotaxis_FCT_PDECO_AT_refactored.py:112-290    (two states, all-time misfit)

The three scripts run the same loop: state solve and adjoint solve for the zero control, then
  1. dk = pointwise gradient expression,
  2. armijo_line_search_ref (helpers.py:1583-1713) -> new states and control,
  3. adjoint solve,  bookkeeping of failed line searches ("fail_count", restarts, backup),
  4. cost functional, relative change as stopping criterion.
Quirk kept on purpose: ``u_backup = uk`` etc. alias the arrays the solvers mutate in place, so
the "restore" at the end only restores the control (np.clip returns a fresh array each time).
"""
from __future__ import annotations

import numpy as np

from . import traj
from .fct import armijo_line_search, cost_functional

DEFAULTS = {
    # script constants: nonlinear_FCT_PDECO_refactored.py:49-65
    "nonlinear": dict(optim="finaltime", beta=1e-1, c_lower=-1.0, c_upper=1.0, tol=1e-4, max_iter_armijo=5,
                      max_iter_GD=50, gam=1e-4, s0=1.0, rescaling=1.0, fail_count_max=3, restart_max=5, min_iters=0),
    # Schnak_FCT_PDECO_refactored.py:54-72
    "schnak": dict(optim="finaltime", beta=1e-1, c_lower=0.0, c_upper=10.0, tol=1e-3, max_iter_armijo=10,
                   max_iter_GD=50, gam=1e-4, s0=1.0, rescaling=1.0, fail_count_max=3, restart_max=5, min_iters=0),
    # chemotaxis_FCT_PDECO_AT_refactored.py:55-75, 136-141 (loop also runs while it < 2, :150)
    "chtxs": dict(optim="alltime", beta=1e-3, c_lower=0.0, c_upper=20.0, tol=1e-4, max_iter_armijo=20,
                  max_iter_GD=50, gam=1e-5, s0=2.0, rescaling=0.1, fail_count_max=5, restart_max=5, min_iters=2),
}


def rel_err(new, old):
    """helpers.py:69-85"""
    return abs(new - old) / abs(old)


def projected_gradient_descent(problem, asm, M, ic, targets, num_steps, dt, wind=None, wind_scale=None, **overrides):
    """Run the loop of the refactored driver ``problem`` in {"nonlinear", "schnak", "chtxs"}.

    ic = (u0,) or (u0, v0); targets = (uhat,) or (uhat, vhat) (final-time vectors or trajectories
    according to ``optim``).  Returns a dict with the final arrays and the per-iteration scalars."""
    P = dict(DEFAULTS[problem])
    P.update(overrides)
    optim, beta, r = P["optim"], P["beta"], P["rescaling"]
    nodes = ic[0].size
    T = num_steps * dt
    vec_length = (num_steps + 1) * nodes
    two = problem != "nonlinear"
    gamma = traj.schnak_params()["gamma"] if problem == "schnak" else None

    def state(c, var1, var2):
        if problem == "nonlinear":
            return traj.solve_nonlinear_equation(c, var1, var2, asm, nodes, num_steps, dt)
        if problem == "schnak":
            return traj.solve_schnak_system(c, var1, var2, asm, nodes, num_steps, dt, wind=wind, wind_scale=wind_scale)
        return traj.solve_chtxs_system(c, var1, var2, asm, nodes, num_steps, dt)

    def adjoint(uk, vk, pk, qk, ck):
        if problem == "nonlinear":
            return traj.solve_adjoint_nonlinear_equation(uk, targets[0], pk, T, asm, nodes, num_steps, dt), None
        if problem == "schnak":
            return traj.solve_adjoint_schnak_system(uk, vk, targets[0], targets[1], pk, qk, T, asm, nodes, num_steps, dt,
                                                    None, optim, wind=wind, wind_scale=wind_scale)
        return traj.solve_adjoint_chtxs_system(uk, vk, targets[0], targets[1], pk, qk, ck, T, asm, nodes, num_steps,
                                               dt, None, optim, rescaling=r)

    def cost(uk, vk, ck):
        if two:
            return cost_functional(uk, targets[0], ck, num_steps, dt, M, beta, optim, var2=vk, var2_target=targets[1])
        return cost_functional(uk, targets[0], ck, num_steps, dt, M, beta, optim)

    ck = np.zeros(vec_length)
    uk = np.zeros(vec_length)
    uk[:nodes] = ic[0]
    vk = None
    if two:
        vk = np.zeros(vec_length)
        vk[:nodes] = ic[1]
    uk, vk = state(ck, uk, vk)
    pk = np.zeros(vec_length)
    qk = np.zeros(vec_length) if two else None
    pk, qk = adjoint(uk, vk, pk, qk, ck)
    cost_fun_old = cost(uk, vk, ck)
    cost_fun_new = (2 + P["tol"]) * cost_fun_old
    stop_crit = rel_err(cost_fun_new, cost_fun_old)

    it = 0
    fail_count = 0
    fail_restart_count = 0
    fail_pass = False
    c_backup, it_backup = ck, 0
    hist = dict(cost=[cost_fun_old], armijo_its=[], stop_crit=[], armijo_margin=[])
    while (stop_crit >= P["tol"] or fail_pass or it < P["min_iters"]) and it < P["max_iter_GD"]:
        if problem == "nonlinear":
            dk = -(beta * ck - pk)
        elif problem == "schnak":
            dk = -(beta * ck - gamma / r * pk)
        else:
            dk = -(beta * ck - qk * uk / r)
        res = armijo_line_search(uk, ck, dk, targets[0], num_steps, dt, P["c_lower"], P["c_upper"], beta, cost_fun_old,
                                 nodes, optim, M, gam=P["gam"], max_iter=P["max_iter_armijo"], s0=P["s0"],
                                 nonlinear_solver=state, var2=vk, var2_target=targets[1] if two else None,
                                 margins=hist["armijo_margin"].append([]) or hist["armijo_margin"][-1])
        if two:
            uk, vk, ck, iters = res
        else:
            uk, ck, iters = res
        pk, qk = adjoint(uk, vk, pk, qk, ck)
        if iters == P["max_iter_armijo"]:
            fail_count += 1
            fail_pass = True
            if it == 0:
                c_backup, it_backup = ck, it
            if fail_count == P["fail_count_max"]:
                break
        else:
            if fail_count > 0:
                fail_count = 0
                fail_restart_count += 1
                fail_pass = False
            if fail_restart_count < P["restart_max"]:
                c_backup, it_backup = ck, it
            elif fail_restart_count == P["restart_max"]:
                break
        cost_fun_new = cost(uk, vk, ck)
        stop_crit = rel_err(cost_fun_new, cost_fun_old)
        hist["cost"].append(cost_fun_new)
        hist["armijo_its"].append(iters)
        hist["stop_crit"].append(stop_crit)
        it += 1
        cost_fun_old = cost_fun_new
    restored = False
    if fail_count == P["fail_count_max"] or fail_restart_count == P["restart_max"] or \
            (it == P["max_iter_GD"] and fail_count > 0):
        ck = c_backup       # the state/adjoint "backups" alias the live arrays (see module docstring)
        restored = True
    return dict(u=uk, v=vk, p=pk, q=qk, c=ck, it=it, it_backup=it_backup, restored=restored, **hist)
