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
        xq, yq = self.ctx.quad_points(mesh.n_cells)
        wx, wy = (wind or rotation_wind(om))(xq, yq)
        self.Arot = self.ctx.assemble_convection(np.stack([wx, wy], axis=1).reshape(-1))

    # -- device-resident API ---------------------------------------------------
    def new_traj(self, batch=None) -> DeviceArray:
        return self.ctx.zeros(self.tlen * (self.batch if batch is None else batch))

    def forward(self, c: DeviceArray, u: DeviceArray, batch=None, c_shared=False):
        """u level 0 holds the IC; fills levels 1..Nt."""
        self.ctx.solidbody_forward(self.Arot, c, u, self.num_steps, self.dt, self.eps, self.rot_scale,
                                   self.drift, self.batch if batch is None else batch, c_shared)

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
