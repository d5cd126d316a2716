#!/usr/bin/env python3
"""Config C2: advection_solidbody_FCT_PDECO_finaltime.py on the MI355X backend.
[-1,1]^2, 81 x 81 P1 nodes, dt = 1e-3, T = 0.25, rotation (-y, x)*40/pi + drift control b = (1,1),
c in [0,5], beta = 1, slotted-disc initial condition, target data/solidbody_t0.25_u.csv (shipped here as
tests/golden/solidbody_t0.25_u.npz), projected gradient descent with the Armijo search of
advection_solidbody_FCT_PDECO_finaltime_Garvie.py:259-317.

usage: python examples/c2_solidbody_pdeco_finaltime.py [--iters 20] [--out results_c2]"""
import argparse
import os
import time

import numpy as np

from _common import ROOT, hp, solvers, slotted_disc, to_dof

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--out", default=None)
args = ap.parse_args()

a1, a2, dx, dt, T = -1.0, 1.0, 0.1 / 2 / 2, 0.001, 0.25
beta, c_lower, c_upper = 1.0, 0.0, 5.0
mesh = hp.SquareMeshP1(a1, a2, round((a2 - a1) / dx))
Nt = round(T / dt)
u0 = to_dof(mesh, slotted_disc(a1, a2, dx))
uhat_T = np.load(os.path.join(ROOT, "tests", "golden", "solidbody_t0.25_u.npz"))["u"]
c0 = np.zeros((Nt + 1) * mesh.nodes)

prob = solvers.SolidBodyDrift(mesh, Nt, dt, om=np.pi / 40, eps=0.0, drift=(1.0, 1.0), order=hp.ORDER_VERTEX)
v2d = mesh.vertex_to_dof
dev = lambda x: hp.reorder_vector_from_dof(x, x.size // mesh.nodes, mesh.nodes, v2d)     # device works in vertex order
t0 = time.perf_counter()
u, p, c, hist = solvers.pgd_solidbody_finaltime(prob, dev(u0), dev(uhat_T), dev(c0), beta, c_lower, c_upper, args.iters,
                                                gam=1e-4, s0=1.0, max_armijo=10, speculative=True)
el = time.perf_counter() - t0
for k, (J, a, s) in enumerate(zip(hist["cost"], hist["armijo_k"], hist["step"])):
    print(f"it {k + 1:3d}  J = {J:.8e}  Armijo trials {a:2d}  step {s:g}")
print(f"{len(hist['cost'])} PGD iterations in {el:.2f} s  ({2 * Nt * (1 + 10) * len(hist['cost']) / el:,.0f} FCT timesteps/s incl. all trials)")
if args.out:
    os.makedirs(args.out, exist_ok=True)
    back = lambda x: hp.reorder_vector_to_dof(x, x.size // mesh.nodes, mesh.nodes, v2d)
    for name, arr in (("u", u), ("p", p), ("c", c)):
        hp.save_trajectory(os.path.join(args.out, f"solidbody_{name}.csv"), back(arr))     # reference CSV layout, DoF order
prob.close()
