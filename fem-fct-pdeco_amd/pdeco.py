"""Projected gradient descent of the reference's refactored PDECO drivers, device resident.

  nonlinear_FCT_PDECO_refactored.py:105-232        "nonlinear"  (one state, final-time misfit)
  Schnak_FCT_PDECO_refactored.py:122-259           "schnak"     (two states, final-time misfit)
  chemotaxis_FCT_PDECO_AT_refactored.py:112-290    "chtxs"      (two states, all-time misfit)

The loop, its constants, the line-search bookkeeping (fail counters, restarts, the control backup)
and the order of the floating-point operations of the pointwise gradient follow the scripts; states,
adjoints, controls and targets stay in HBM and only scalars (costs, norms) reach the host.
``speculative=True`` evaluates all Armijo trial steps s0/2^k of an iteration as one batch of
independent trajectories (helpers.py:1681-1708 picks the first accepted k: same iterate, the states
agree to the solver tolerance) -- the lever that fills the GPU on these small meshes.
"""
from __future__ import annotations

import time

import numpy as np

from . import _lib
from .mesh import SquareMeshP1
from .systems import PDESystems, _chtxs_par, _schnak_par, get_nonlinear_eqns_params

DEFAULTS = {
    # nonlinear_FCT_PDECO_refactored.py:49-65
    "nonlinear": dict(optim="finaltime", beta=1e-1, c_lower=-1.0, c_upper=1.0, tol=1e-4, max_iter_armijo=5,
                      max_iter_GD=50, gam=1e-4, s0=1.0, rescaling=1.0, fail_count_max=3, restart_max=5, min_iters=0),
    # Schnak_FCT_PDECO_refactored.py:54-72
    "schnak": dict(optim="finaltime", beta=1e-1, c_lower=0.0, c_upper=10.0, tol=1e-3, max_iter_armijo=10,
                   max_iter_GD=50, gam=1e-4, s0=1.0, rescaling=1.0, fail_count_max=3, restart_max=5, min_iters=0),
    # chemotaxis_FCT_PDECO_AT_refactored.py:55-75, 136-141, 150
    "chtxs": dict(optim="alltime", beta=1e-3, c_lower=0.0, c_upper=20.0, tol=1e-4, max_iter_armijo=20,
                  max_iter_GD=50, gam=1e-5, s0=2.0, rescaling=0.1, fail_count_max=5, restart_max=5, min_iters=2),
}


def rel_err(new, old):
    """helpers.py:69-85"""
    return abs(new - old) / abs(old)


class SystemPDECO:
    """One optimisation problem on one GPU.  ``V``: mesh descriptor (stands in for the dolfin
    FunctionSpace); host vectors are in FEniCS DoF order like the reference's."""

    def __init__(self, problem: str, V: SquareMeshP1, num_steps: int, dt: float, device_id: int = 0, wind=None,
                 wind_scale=None, **overrides):
        """``wind`` / ``wind_scale`` (problem "schnak" only): the separable time-dependent wind ``s(t) w0(x)`` of the
        script BASELINE config 3 names (Schnak_FCT_PDECO_alltime.py:55,174-175), see systems.solve_schnak_system."""
        if problem not in DEFAULTS:
            raise ValueError(f"unknown problem '{problem}' (one of {sorted(DEFAULTS)})")
        if (wind is not None or wind_scale is not None) and problem != "schnak":
            raise ValueError("wind / wind_scale: only the Schnakenberg driver has a time-dependent wind")
        self.problem, self.V, self.Nt, self.dt = problem, V, int(num_steps), float(dt)
        self.P = dict(DEFAULTS[problem])
        unknown = set(overrides) - set(self.P)
        if unknown:
            raise TypeError(f"unknown option(s) {sorted(unknown)}")
        self.P.update(overrides)
        if self.P["optim"] not in ("alltime", "finaltime"):
            raise ValueError(f"Invalid value for 'optim': '{self.P['optim']}'. Must be one of ['alltime', 'finaltime'].")
        self.S = PDESystems(V, device_id=device_id, order=_lib.ORDER_VERTEX)
        self.ctx, self.n = self.S.ctx, self.S.n
        self.tl = (self.Nt + 1) * self.n
        self.v2d = np.asarray(V.vertex_to_dof, dtype=np.int64)
        self.two = problem != "nonlinear"
        if problem == "nonlinear":
            self.eps, _, wind = get_nonlinear_eqns_params()
            self.Aw, self.AwT = self.S.convection(wind, "nonlinear")
        elif problem == "schnak":
            from .systems import _wind_factors
            self.par, wind0 = _schnak_par()
            self.Aw, self.AwT = self.S.convection(wind or wind0, None if wind is not None else "schnak")
            self.wscale = _wind_factors(wind_scale, self.Nt, self.dt)                               # forward: t += dt
            self.wscale_adj = _wind_factors(wind_scale, self.Nt, self.dt, T=self.Nt * self.dt)      # adjoint: t = T; t -= dt
        else:
            self.par = _chtxs_par()
        self._arrays = []

    # ------------------------------------------------------------------ staging
    def _up(self, x, count=None):
        x = np.asarray(x, dtype=np.float64).ravel()
        d = self.ctx.array(np.ascontiguousarray(x.reshape(-1, self.n)[:, self.v2d]).ravel())
        self._arrays.append(d)
        return d

    def _zeros(self, count):
        d = self.ctx.zeros(count)
        self._arrays.append(d)
        return d

    def _down(self, d, count=None, off=0):
        count = d.count if count is None else count
        tmp = self.ctx.empty(count)
        try:
            tmp.copy_from(d, count, src_off=off)
            h = tmp.download().reshape(-1, self.n)
        finally:
            tmp.free()
        out = np.empty_like(h)
        out[:, self.v2d] = h
        return out.ravel()

    def close(self):
        for a in self._arrays:
            a.free()
        self._arrays = []
        self.S.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ solves (batch of B trajectories)
    def _state(self, c, u, v, clev, B):
        """solve_<problem>(c, u, v): the control is frozen at time level 1 (helpers.py:577-578 etc.)"""
        n, tl = self.n, self.tl
        for b in range(B):
            clev.copy_from(c, n, dst_off=b * n, src_off=b * tl + n)
        if self.problem == "nonlinear":
            self.ctx.nonlinear_forward(self.Aw, clev, u, self.Nt, self.dt, self.eps, batch=B)
        elif self.problem == "schnak":
            # the drivers and armijo_line_search_ref call the state solver without `rescaling`
            # (helpers.py:1685): its defaults apply, 1 (helpers.py:512) and 1/10 (helpers.py:1252)
            self.ctx.schnak_forward(self.Aw, clev, u, v, self.Nt, self.dt, self.par, 1.0, batch=B, wind_scale=self.wscale)
        else:
            self.ctx.chtxs_forward(clev, u, v, self.Nt, self.dt, self.par, 0.1, batch=B)

    def _adjoint(self, u, v, p, q, c, tg):
        if self.problem == "nonlinear":
            self.ctx.nonlinear_adjoint(self.Aw, u, tg[0], p, self.Nt, self.dt, self.eps)
        elif self.problem == "schnak":
            self.ctx.schnak_adjoint(self.AwT, u, v, tg[0], tg[1], p, q, self.Nt, self.dt, self.par,
                                    alltime=self.P["optim"] == "alltime", wind_scale=self.wscale_adj)
        else:
            self.ctx.chtxs_adjoint(u, v, tg[0], tg[1], p, q, c, self.Nt, self.dt, self.par, self.P["rescaling"],
                                   self.P["optim"] == "alltime")

    def _cost(self, u, v, c, tg, B=1):
        return self.ctx.cost_functional(u, tg[0], c, self.Nt, self.dt, self.P["beta"], self.P["optim"],
                                        var2=v if self.two else None, var2_target=tg[1] if self.two else None, batch=B)

    def _descent(self, c, u, p, q, d):
        beta, r = self.P["beta"], self.P["rescaling"]
        if self.problem == "nonlinear":
            self.ctx.descent_pointwise(self.tl, beta, c, p, d)                       # -(beta*ck - pk)
        elif self.problem == "schnak":
            self.ctx.descent_pointwise(self.tl, beta, c, p, d, scale=self.par[3] / r)   # -(beta*ck - gamma/r*pk)
        else:
            self.ctx.descent_pointwise(self.tl, beta, c, q, d, y=u, divisor=r)       # -(beta*ck - qk*uk/r)

    # ------------------------------------------------------------------ the loop
    def run(self, ic, targets, speculative=True, callback=None):
        """ic = (u0,) / (u0, v0); targets = (uhat,) / (uhat, vhat): final-time vectors (n) or trajectories
        ((Nt+1)*n) according to ``optim``.  Returns a dict: final u, v, p, q, c (NumPy, FEniCS order),
        ``cost`` (initial value first), ``armijo_its``, ``stop_crit``, ``it``, ``restored``."""
        P, ctx, n, tl, Nt, dt = self.P, self.ctx, self.n, self.tl, self.Nt, self.dt
        K = int(P["max_iter_armijo"])
        B = K if speculative else 1
        tsz = tl if P["optim"] == "alltime" else n
        if len(ic) != (2 if self.two else 1) or len(targets) != len(ic):
            raise ValueError("ic / targets: one entry per state variable")
        for t in targets:
            if np.asarray(t).size != tsz:
                raise ValueError(f"target of {np.asarray(t).size} values, expected {tsz} for optim='{P['optim']}'")

        def replicate(one, count, reps):
            """reps copies of a device vector, made on the device"""
            d = self._zeros(reps * count)
            for b in range(reps):
                d.copy_from(one, count, dst_off=b * count)
            return d

        def traj0(x0, reps=1):
            a = np.zeros(tl)
            a[:n] = np.asarray(x0, dtype=np.float64)
            one = self._up(a)
            return one if reps == 1 else replicate(one, tl, reps)

        u = traj0(ic[0])
        v = traj0(ic[1]) if self.two else None
        p = self._zeros(tl)
        q = self._zeros(tl) if self.two else None
        c, d, cbak = self._zeros(tl), self._zeros(tl), self._zeros(tl)
        tg = [self._up(t) for t in targets]
        clev = self._zeros(B * n)
        uB = traj0(ic[0], B)
        vB = traj0(ic[1], B) if self.two else None
        cB = self._zeros(B * tl)
        if speculative:
            ckB = self._zeros(B * tl)
            tgB = [replicate(t, tsz, B) for t in tg]
        else:
            ckB, tgB = c, tg

        self._state(c, u, v, clev, 1)
        self._adjoint(u, v, p, q, c, tg)
        cost_old = float(self._cost(u, v, c, tg)[0])
        cost_new = (2 + P["tol"]) * cost_old
        stop_crit = rel_err(cost_new, cost_old)
        it = fail_count = fail_restart_count = 0
        fail_pass = False
        it_backup = 0
        # armijo_margin[it][k]: distance of trial k's Armijo test from its threshold relative to the cost,
        # (J_trial - J_k + gam/s ||c_inc - c||^2_Q) / |J_k| (> 0: rejected), for the trials the sequential search looks at
        # wall[it]: host clock after iteration it's cost evaluation (a read-back: the device has finished the iteration);
        # wall0: after the initial state + adjoint + cost -- so (wall[-1] - wall0) / it is the time of an iteration proper
        hist = dict(cost=[cost_old], armijo_its=[], stop_crit=[], armijo_margin=[], wall=[], wall0=time.perf_counter())
        svals = [P["s0"] / 2 ** k for k in range(K)]
        while (stop_crit >= P["tol"] or fail_pass or it < P["min_iters"]) and it < P["max_iter_GD"]:
            self._descent(c, u, p, q, d)
            # armijo_line_search_ref, helpers.py:1681-1713
            if speculative:
                for k, s in enumerate(svals):
                    ctx.project_control(c, s, d, P["c_lower"], P["c_upper"], cB.ptr + 8 * k * tl, tl)
                    ckB.copy_from(c, tl, dst_off=k * tl)
                self._state(cB, uB, vB, clev, B)
                J = self._cost(uB, vB, cB, tgB, B)
                dif = ctx.l2_norm_sq_Q(cB, ckB, Nt, dt, batch=B)
                acc = K - 1
                margins = []
                for k, s in enumerate(svals):
                    margins.append((float(J[k]) - cost_old + P["gam"] / s * float(dif[k])) / abs(cost_old))
                    if J[k] - cost_old <= -P["gam"] / s * dif[k]:
                        acc = k
                        break
            else:
                acc = K - 1
                margins = []
                for k, s in enumerate(svals):
                    ctx.project_control(c, s, d, P["c_lower"], P["c_upper"], cB, tl)
                    self._state(cB, uB, vB, clev, 1)
                    Jk = float(self._cost(uB, vB, cB, tg)[0])
                    dif = float(ctx.l2_norm_sq_Q(cB, c, Nt, dt)[0])
                    margins.append((Jk - cost_old + P["gam"] / s * dif) / abs(cost_old))
                    if Jk - cost_old <= -P["gam"] / s * dif:
                        acc = k
                        break
            hist["armijo_margin"].append(margins)
            slot = acc if speculative else 0
            c.copy_from(cB, tl, src_off=slot * tl)
            u.copy_from(uB, tl, src_off=slot * tl)
            if self.two:
                v.copy_from(vB, tl, src_off=slot * tl)
            iters = acc + 1
            self._adjoint(u, v, p, q, c, tg)
            # line-search bookkeeping (e.g. Schnak_FCT_PDECO_refactored.py:180-215)
            if iters == K:
                fail_count += 1
                fail_pass = True
                if it == 0:
                    cbak.copy_from(c, tl)
                    it_backup = it
                if fail_count == P["fail_count_max"]:
                    break
            else:
                if fail_count > 0:
                    fail_count = 0
                    fail_restart_count += 1
                    fail_pass = False
                if fail_restart_count < P["restart_max"]:
                    cbak.copy_from(c, tl)
                    it_backup = it
                elif fail_restart_count == P["restart_max"]:
                    break
            cost_new = float(self._cost(u, v, c, tg)[0])
            stop_crit = rel_err(cost_new, cost_old)
            hist["cost"].append(cost_new)
            hist["armijo_its"].append(iters)
            hist["stop_crit"].append(stop_crit)
            hist["wall"].append(time.perf_counter())
            if callback is not None:
                callback(it, cost_new, iters, stop_crit)
            it += 1
            cost_old = cost_new
        restored = False
        if fail_count == P["fail_count_max"] or fail_restart_count == P["restart_max"] or \
                (it == P["max_iter_GD"] and fail_count > 0):
            # the scripts' state/adjoint "backups" alias the live arrays: only the control is restored
            c.copy_from(cbak, tl)
            restored = True
        out = dict(u=self._down(u), v=self._down(v) if self.two else None, p=self._down(p),
                   q=self._down(q) if self.two else None, c=self._down(c), it=it, it_backup=it_backup,
                   restored=restored, **hist)
        for a in self._arrays:
            a.free()
        self._arrays = []
        return out


def projected_gradient_descent(problem, V, ic, targets, num_steps, dt, speculative=True, device_id=0, wind=None,
                               wind_scale=None, **overrides):
    """One call = one run of the refactored driver ``problem`` (see module docstring)."""
    with SystemPDECO(problem, V, num_steps, dt, device_id=device_id, wind=wind, wind_scale=wind_scale, **overrides) as prob:
        return prob.run(ic, targets, speculative=speculative)
