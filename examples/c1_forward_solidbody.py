#!/usr/bin/env python3
"""Config C1 (plumbing): the forward run of advection_solidbody_FCT.py -- rotation 40/pi + constant drift (2,2),
slotted disc of radius 1/3 with a slit of 0.1, [-1,1]^2, 81 x 81, dt = 1e-3 -- twice: as a device-resident sweep, and
step by step through the drop-in operator with the old sign convention (`FCT_alg(A, rhs, u_n, dt, nodes, M, M_Lump,
dof_neighbors)`, SciPy matrices in, NumPy out) exactly as the script's time loop does (advection_solidbody_FCT.py:129-148).

usage: python examples/c1_forward_solidbody.py [--steps 50]"""
import argparse
import time

import numpy as np

from _common import hp, solvers, slotted_disc, to_dof

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
args = ap.parse_args()
a1, a2, dx, dt = -1.0, 1.0, 0.025, 0.001
mesh = hp.SquareMeshP1(a1, a2, round((a2 - a1) / dx))
n, Nt = mesh.nodes, args.steps
u0 = to_dof(mesh, slotted_disc(a1, a2, dx, slit=0.1))

# --- device-resident sweep: the constant "control" 1 with drift (2,2) is dot(drift, grad(v))*u*dx
prob = solvers.SolidBodyDrift(mesh, Nt, dt, om=np.pi / 40, eps=0.0, drift=(2.0, 2.0), order=hp.ORDER_FENICS)
uk = np.zeros((Nt + 1) * n)
uk[:n] = u0
t0 = time.perf_counter()
prob.solve_state(np.ones((Nt + 1) * n), uk)
t_dev = time.perf_counter() - t0
prob.close()

# --- the same steps through the drop-in operator, matrices as SciPy CSR in FEniCS DoF order
S = hp.systems._system(mesh)
M = hp.assemble_mass(mesh)
M_lumped = hp.row_lump(M, n)
wind = lambda x, y: (-y * 40 / np.pi + 2.0, x * 40 / np.pi + 2.0)              # rotation + drift
A = hp.systems.device_matrix(mesh, S.convection(wind, "c1")[0])                 # assemble_sparse(dot(wind, grad(v))*u*dx)
u = u0.copy()
t0 = time.perf_counter()
for _ in range(Nt):
    u = hp.FCT_alg(A, np.zeros(n), u, dt, n, M, M_lumped, None)
t_host = time.perf_counter() - t0
ml = M_lumped.diagonal()
print(f"device-resident sweep : {Nt} steps in {t_dev * 1e3:8.1f} ms")
print(f"drop-in FCT_alg loop  : {Nt} steps in {t_host * 1e3:8.1f} ms (matrix upload + download every call)")
print(f"difference of the two end states: {np.abs(u - uk[Nt * n:]).max():.2e}")
print(f"mass {u0 @ ml:.8f} -> {u @ ml:.8f},  min {u.min():+.2e}, max {u.max():.6f}")
