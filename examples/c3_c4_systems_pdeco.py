#!/usr/bin/env python3
"""Configs C3 / C4: the refactored PDECO drivers of the Schnakenberg and chemotaxis systems on the MI355X backend
(UnitSquare, 41 x 41 P1 nodes, dt = 5e-4, T = 0.1; script constants of Schnak_FCT_PDECO_refactored.py and
chemotaxis_FCT_PDECO_AT_refactored.py).  Targets: the build's own forward solve at the true control, as the
reference workflow does (chemotaxis_generate_pattern_FCT.py:90-96).

usage: python examples/c3_c4_systems_pdeco.py {schnak,chtxs,nonlinear} [--iters 5] [--optim alltime|finaltime]"""
import argparse
import time

import numpy as np

from _common import hp, pdeco

ap = argparse.ArgumentParser()
ap.add_argument("problem", choices=["schnak", "chtxs", "nonlinear"])
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--optim", default=None)
args = ap.parse_args()

dx, dt, T = 0.025, 5e-4, 0.1
V = hp.SquareMeshP1(0.0, 1.0, round(1 / dx))
n, Nt = V.nodes, round(T / dt)
tl = (Nt + 1) * n
z = lambda x0: np.concatenate([x0, np.zeros(Nt * n)])
if args.problem == "schnak":
    ic = hp.schnak_sys_IC(0, 1, dx, n, V.vertex_to_dof)
    full = hp.solve_schnak_system(np.full(tl, 0.1), z(ic[0]), z(ic[1]), V, n, Nt, dt, None)          # true control a = 0.1
elif args.problem == "chtxs":
    ic = hp.chtxs_sys_IC(0, 1, dx, n, V.vertex_to_dof)
    full = hp.solve_chtxs_system(np.full(tl, 10.0), z(ic[0]), z(ic[1]), V, n, Nt, dt, None)           # c = 100 * r, r = 1/10
else:
    ic = (hp.nonlinear_equation_IC(0, 1, dx, n, V.vertex_to_dof),)
    full = (hp.solve_nonlinear_equation(np.full(tl, 0.5), z(ic[0]), None, V, n, Nt, dt, None)[0],)
opts = dict(max_iter_GD=args.iters, tol=0.0)
if args.optim:
    opts["optim"] = args.optim
optim = opts.get("optim", pdeco.DEFAULTS[args.problem]["optim"])
targets = tuple(np.array(f) if optim == "alltime" else np.array(f[Nt * n:]) for f in full)
t0 = time.perf_counter()
res = hp.projected_gradient_descent(args.problem, V, ic, targets, Nt, dt, speculative=True, **opts)
el = time.perf_counter() - t0
print(f"{args.problem} ({optim}): {res['it']} PGD iterations in {el:.2f} s, restored = {res['restored']}")
for k, J in enumerate(res["cost"]):
    trials = res["armijo_its"][k - 1] if k else "-"
    print(f"  it {k:2d}  J = {J:.8e}   Armijo trials {trials}")
