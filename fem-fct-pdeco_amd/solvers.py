"""Trajectory solvers: forward and adjoint sweeps resident on the GPU.

Mirrors the time loops the reference writes in Python around ``FCT_alg[_ref]``
(one host<->device crossing per sweep instead of ~10 per time step):

  solid-body rotation + drift control (inline loops of the advection scripts)
      forward   advection_solidbody_FCT_PDECO_finaltime.py:175-193
      adjoint   advection_solidbody_FCT_PDECO_finaltime.py:200-221,
                advection_solidbody_FCT_PDECO_alltime.py:232-259
      gradient  advection_solidbody_FCT_PDECO_finaltime.py:228-238
      Armijo    advection_solidbody_FCT_PDECO_finaltime_Garvie.py:259-317

NumPy-facing methods take/return ``(num_steps+1)*nodes`` float64 arrays in FEniCS
DoF order, mutate the state array in place *and* return it, like the reference.
"""
from __future__ import annotations

import time

import numpy as np

from . import _lib
from .device import Context, DeviceArray
from .mesh import SquareMeshP1


def rotation_wind(om):
    """``1/om * Expression(('-x[1]','x[0]'))`` (advection_solidbody_FCT_PDECO_finaltime.py:91-93)."""
    return lambda x, y: (-(1.0 / om) * y, (1.0 / om) * x)


class SolidBodyDrift:
    """Drift-control advection problem on one GPU; ``batch`` independent trajectories advance
    together in every kernel launch (Armijo trial steps, regularisation sweeps)."""

    def __init__(self, mesh: SquareMeshP1, num_steps: int, dt: float, om=np.pi / 40, eps=0.0,
                 drift=(1.0, 1.0), rot_scale=1.0, batch=1, device_id=0, order=_lib.ORDER_FENICS,
                 wind=None):
        self.mesh, self.num_steps, self.dt = mesh, int(num_steps), float(dt)
        self.eps, self.drift, self.rot_scale, self.batch = float(eps), tuple(map(float, drift)), float(rot_scale), int(batch)
        self.order = order
        self.ctx = Context(device_id)
        self.ctx.set_mesh_square(mesh.a1, mesh.a2, mesh.n_cells, order)
        self.n = self.ctx.n
        self.tlen = (self.num_steps + 1) * self.n
        if wind is None:        # 1/om * (-x[1], x[0]) (finaltime.py:91-93): linear, assembled in closed form
            self.Arot = self.ctx.assemble_rotation(1.0 / om)
        else:
            xq, yq = self.ctx.quad_points(mesh.n_cells)
            wx, wy = wind(xq, yq)
            self.Arot = self.ctx.assemble_convection(np.stack([wx, wy], axis=1).reshape(-1))

    # -- device-resident API ---------------------------------------------------
    def new_traj(self, batch=None) -> DeviceArray:
        return self.ctx.zeros(self.tlen * (self.batch if batch is None else batch))

    def forward(self, c: DeviceArray, u: DeviceArray, batch=None, c_shared=False, src=None):
        """u level 0 holds the IC; fills levels 1..Nt.  ``src``: optional source trajectory, the step to
        level n+1 gets rhs = assemble(src_{n+1}*v*dx) (advection_FCT_PDECO_alltime_exact.py:249-253)."""
        self.ctx.solidbody_forward(self.Arot, c, u, self.num_steps, self.dt, self.eps, self.rot_scale,
                                   self.drift, self.batch if batch is None else batch, c_shared, src_traj=src)

    def adjoint(self, c, u, uhat, p, optim="finaltime", batch=None, c_shared=False):
        if optim not in ("alltime", "finaltime"):
            raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of ['alltime', 'finaltime'].")
        self.ctx.solidbody_adjoint(self.Arot, c, u, uhat, p, self.num_steps, self.dt, self.eps, self.rot_scale,
                                   self.drift, optim == "alltime", self.batch if batch is None else batch, c_shared)

    def descent_direction(self, c, u, p, beta, d, scratch=None):
        """d_k = ChebSI(-(beta*M*c_k + int p_k (b.grad u_k) v)) for every level k (finaltime.py:228-238);
        all levels are one batched launch sequence."""
        levels = self.num_steps + 1
        own = scratch is None
        rhs = self.ctx.empty(self.tlen) if own else scratch
        try:
            self.ctx.drift_gradient_rhs(c, u, p, beta, rhs, levels, self.drift)
            self.ctx.chebsi(rhs, d, 20, 0.5, 2.0, batch=levels)
        finally:
            if own:
                rhs.free()

    def cost(self, u, target, c, beta, optim, batch=None):
        return self.ctx.cost_functional(u, target, c, self.num_steps, self.dt, beta, optim,
                                        batch=self.batch if batch is None else batch)

    # -- NumPy-facing mirrors of the inline reference loops --------------------------
    def solve_state(self, ck, uk):
        """finaltime.py:175-193: mutates ``uk[nodes:]`` in place and returns ``uk``."""
        c = self.ctx.array(ck)
        u = self.ctx.array(uk)
        try:
            self.forward(c, u, batch=1)
            u.download(uk)
        finally:
            c.free()
            u.free()
        return uk

    def solve_adjoint(self, ck, uk, uhat, pk, optim="finaltime"):
        """finaltime.py:200-221 / alltime.py:232-259: fills and returns ``pk``."""
        c = self.ctx.array(ck)
        u = self.ctx.array(uk)
        uh = self.ctx.array(uhat)
        p = self.ctx.zeros(self.tlen)
        try:
            self.adjoint(c, u, uh, p, optim, batch=1)
            p.download(pk)
        finally:
            for a in (c, u, uh, p):
                a.free()
        return pk

    def solve_descent_direction(self, ck, uk, pk, beta):
        c, u, p = self.ctx.array(ck), self.ctx.array(uk), self.ctx.array(pk)
        d = self.ctx.empty(self.tlen)
        try:
            self.descent_direction(c, u, p, beta, d)
            return d.download()
        finally:
            for a in (c, u, p, d):
                a.free()

    def solver_log(self, batch=None):
        return self.ctx.traj_info(self.num_steps, self.batch if batch is None else batch)

    def close(self):
        self.ctx.close()


class LinearSourceControl(SolidBodyDrift):
    """Linear advection-diffusion with a distributed source control, the problem family of
    advection_FCT_PDECO_{alltime,finaltime}[_exact].py:  A_u = A - eps*Ad (state), A_p = -A - eps*Ad (adjoint),
    state rhs assemble((g + c)*v*dx), adjoint rhs assemble((uhat - u)*v*dx) (all-time) or terminal condition
    uhat_T - u(T) (final-time).  ``wind(x, y) -> (wx, wy)`` as in the scripts' ``velocity``."""

    def __init__(self, mesh, num_steps, dt, wind, eps=1e-3, batch=1, device_id=0, order=_lib.ORDER_FENICS):
        super().__init__(mesh, num_steps, dt, eps=eps, drift=(0.0, 0.0), rot_scale=1.0, batch=batch,
                         device_id=device_id, order=order, wind=wind)
        self._zero_c = self.ctx.zeros(self.tlen)     # no drift control: the control enters through the source

    def state(self, src: DeviceArray, u: DeviceArray, batch=None):
        """advection_FCT_PDECO_alltime_exact.py:236-253 (src = g + c, both trajectories)"""
        self.forward(self._zero_c, u, batch=batch, c_shared=True, src=src)

    def adjoint_state(self, u, uhat, p, optim="alltime", batch=None):
        """:259-274 (all-time) / advection_FCT_PDECO_finaltime.py (final-time)"""
        self.adjoint(self._zero_c, u, uhat, p, optim, batch=batch, c_shared=True)

    def solve_state(self, src, uk):
        s, u = self.ctx.array(src), self.ctx.array(uk)
        try:
            self.state(s, u, batch=1)
            u.download(uk)
        finally:
            s.free()
            u.free()
        return uk

    def solve_adjoint_state(self, uk, uhat, pk, optim="alltime"):
        u, uh, p = self.ctx.array(uk), self.ctx.array(uhat), self.ctx.zeros(self.tlen)
        try:
            self.adjoint_state(u, uh, p, optim, batch=1)
            p.download(pk)
        finally:
            for a in (u, uh, p):
                a.free()
        return pk


def pgd_solidbody_finaltime(prob: SolidBodyDrift, u0, uhat_T, c0, beta, c_lower, c_upper, iters,
                            gam=1e-4, s0=1.0, max_armijo=10, speculative=True, tol=None):
    """Final-time variant of :func:`pgd_solidbody` (advection_solidbody_FCT_PDECO_finaltime_Garvie.py)."""
    return pgd_solidbody(prob, u0, uhat_T, c0, beta, c_lower, c_upper, iters, gam, s0, max_armijo, speculative, tol,
                         optim="finaltime")


def pgd_solidbody_alltime(prob: SolidBodyDrift, u0, uhat_all, c0, beta, c_lower, c_upper, iters,
                          gam=1e-4, s0=1.0, max_armijo=10, speculative=True, tol=None):
    """All-time variant (advection_solidbody_FCT_PDECO_alltime_Garvie.py, config C5's loop): ``uhat_all``
    is the target trajectory ((Nt+1)*n, level 0 = u0)."""
    return pgd_solidbody(prob, u0, uhat_all, c0, beta, c_lower, c_upper, iters, gam, s0, max_armijo, speculative, tol,
                         optim="alltime")


def pgd_solidbody(prob: SolidBodyDrift, u0, uhat, c0, beta, c_lower, c_upper, iters,
                  gam=1e-4, s0=1.0, max_armijo=10, speculative=True, tol=None, optim="finaltime"):
    """Projected gradient descent for the drift-control problem, following the loop of
    advection_solidbody_FCT_PDECO_finaltime_Garvie.py:164-330 / ..._alltime_Garvie.py:164-340 step for step:

        adjoint(c_prev, u) -> d = ChebSI(-(beta M c_prev + int p (b.grad u) v)) -> c = clip(c_prev + s0 d)
        -> state(c) -> J_k -> Armijo: trials c_inc = clip(c + s d), s = s0/2^k, accept the first with
        J(c_inc) - J_k <= -gam/s ||c_inc - c||^2_Q  (else the last) -> c_prev = c_inc

    optim="finaltime": target uhat(T) (n values), p(T) = uhat_T - u(T), u(T) seeded with the target before
    the first adjoint solve; optim="alltime": target trajectory, p(T) = 0, misfit load at every level,
    u seeded with the whole target trajectory (alltime_Garvie.py: ``uk = np.copy(uhat_all)``).
    Everything stays in HBM; the host sees scalars only.  ``speculative=True`` evaluates all
    ``max_armijo`` trial steps as one batch of independent trajectories (same launches, B = max_armijo)
    and picks the first accepted one -- the iterate of the sequential search (each trial trajectory is
    the same computation; the states agree to the low-order solver tolerance, 1e-13).
    Returns ``(u, p, c, history)`` as NumPy arrays + a dict of per-iteration scalars."""
    if optim not in ("alltime", "finaltime"):
        raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of ['alltime', 'finaltime'].")
    alltime = optim == "alltime"
    ctx, n, Nt, dt, tl = prob.ctx, prob.n, prob.num_steps, prob.dt, prob.tlen
    B = int(max_armijo) if speculative else 1
    u = ctx.zeros(tl)
    u.upload(np.concatenate([np.asarray(u0, dtype=np.float64), np.zeros(tl - n)]))
    # the reference seeds u(T) with the target before the first adjoint solve (finaltime.py:146)
    p, d, c, rhs = ctx.zeros(tl), ctx.zeros(tl), ctx.zeros(tl), ctx.empty(tl)
    c_prev = ctx.array(np.asarray(c0, dtype=np.float64))
    uhat = np.asarray(uhat, dtype=np.float64).ravel()
    if uhat.size != (tl if alltime else n):
        raise ValueError(f"target of {uhat.size} values, expected {tl if alltime else n} for optim='{optim}'")
    uh = ctx.array(uhat)
    uhB = ctx.zeros(B * uhat.size)                      # B copies of the target, replicated on the device
    for k in range(B):
        uhB.copy_from(uh, uhat.size, dst_off=k * uhat.size)
    cB, uB, ckB = ctx.zeros(B * tl), ctx.zeros(B * tl), ctx.zeros(B * tl)
    for k in range(B):                                  # level 0 of every trial trajectory = the initial condition
        uB.copy_from(u, n, dst_off=k * tl)
    # armijo_margin: per iteration, for every trial the sequential search looks at, the distance of the Armijo test from
    # its threshold relative to the cost, (J_trial - J_k + gam/s ||c_inc - c||^2_Q) / |J_k|  (> 0: rejected).  A margin of
    # the size of the solver tolerance would mean that two faithful implementations may decide differently.
    hist = dict(cost=[], armijo_k=[], step=[], rel_change=[], armijo_margin=[], wall=[], wall0=time.perf_counter())
    if alltime:
        u.copy_from(uh, tl - n, dst_off=n, src_off=n)      # uk = np.copy(uhat_all), level 0 = u0
    else:
        u.copy_from(uh, n, dst_off=Nt * n)                 # uk[num_steps*nodes:] = uhat_T
    try:
        for it in range(iters):
            prob.adjoint(c_prev, u, uh, p, optim, batch=1)
            prob.descent_direction(c_prev, u, p, beta, d, scratch=rhs)
            ctx.project_control(c_prev, s0, d, c_lower, c_upper, c, tl)
            prob.forward(c, u, batch=1)
            J_k = float(prob.cost(u, uh, c, beta, optim, batch=1)[0])
            svals = [s0 * (1 / 2 ** k) for k in range(max_armijo)]
            accepted = None
            if speculative:
                for k, s in enumerate(svals):
                    ctx.project_control(c, s, d, c_lower, c_upper, cB.ptr + 8 * k * tl, tl)
                    ckB.copy_from(c, tl, dst_off=k * tl)
                prob.forward(cB, uB, batch=B)
                J = prob.cost(uB, uhB, cB, beta, optim, batch=B)
                stat = ctx.l2_norm_sq_Q(cB, ckB, Nt, dt, batch=B)
                margins = []
                for k, s in enumerate(svals):
                    accepted = k
                    margins.append((float(J[k]) - J_k + gam / s * float(stat[k])) / abs(J_k))
                    if not (J[k] - J_k > -gam / s * stat[k]):
                        break
                J_acc = float(J[accepted])
                c_prev.copy_from(cB, tl, src_off=accepted * tl)
                u.copy_from(uB, tl, src_off=accepted * tl)
            else:
                margins = []
                for k, s in enumerate(svals):
                    accepted = k
                    ctx.project_control(c, s, d, c_lower, c_upper, cB, tl)
                    prob.forward(cB, uB, batch=1)
                    J_acc = float(prob.cost(uB, uh, cB, beta, optim, batch=1)[0])
                    stat = float(ctx.l2_norm_sq_Q(cB, c, Nt, dt)[0])
                    margins.append((J_acc - J_k + gam / s * stat) / abs(J_k))
                    if not (J_acc - J_k > -gam / s * stat):
                        break
                c_prev.copy_from(cB, tl)
                u.copy_from(uB, tl)
            hist["cost"].append(J_acc)
            hist["wall"].append(time.perf_counter())          # (J_acc is a read-back: the iteration's device work is done)
            hist["armijo_margin"].append(margins)
            hist["armijo_k"].append(accepted + 1)
            hist["step"].append(svals[accepted])
            hist["rel_change"].append(abs(J_k - J_acc) / abs(J_k))
            if tol is not None and hist["rel_change"][-1] < tol:
                break
        hist["armijo_margin_min"] = min((abs(m) for ms in hist["armijo_margin"] for m in ms), default=None)
        return u.download(), p.download(), c_prev.download(), hist
    finally:
        for a in (u, p, d, c, rhs, c_prev, uh, uhB, cB, uB, ckB):
            a.free()
