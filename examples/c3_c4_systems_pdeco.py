#!/usr/bin/env python3
"""Configs C3 / C4: the refactored PDECO drivers of the Schnakenberg and chemotaxis systems on the MI355X backend
(UnitSquare, 41 x 41 P1 nodes, dt = 5e-4, T = 0.1; script constants of Schnak_FCT_PDECO_refactored.py and
chemotaxis_FCT_PDECO_AT_refactored.py).  Targets: the build's own forward solve at the true control, as the
reference workflow does (chemotaxis_generate_pattern_FCT.py:90-96).

`--named-c3` (with `schnak`): BASELINE config 3 as the script it names sets the problem up (Schnak_FCT_PDECO_alltime.py:22-55,
174-175): all-time misfit, control box [0, 0.5], wind (-(y-.5), (x-.5)) * sin(2 pi t) re-assembled per step (a per-level
factor on the device), on BASELINE's grid dx = 0.025, dt = 5e-4 (the script's own literals, dx = 0.02 / dt = 2e-3, put the
omega1-scaled convection at CFL 7: far outside the scheme's dt restriction, where the low-order solve falls back to
BiCGStab and the adjoint operator, with its indefinite reaction matrix, is no longer solved to 1e-13).

usage: python examples/c3_c4_systems_pdeco.py {schnak,chtxs,nonlinear} [--iters 5] [--optim alltime|finaltime] [--named-c3]"""
import argparse
import time

import numpy as np

from _common import hp, pdeco

ap = argparse.ArgumentParser()
ap.add_argument("problem", choices=["schnak", "chtxs", "nonlinear"])
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--optim", default=None)
ap.add_argument("--named-c3", action="store_true")
args = ap.parse_args()

dx, dt, T = 0.025, 5e-4, 0.1
wind_kw = {}
if args.named_c3:
    if args.problem != "schnak":
        raise SystemExit("--named-c3 goes with the schnak problem")
    wind_kw = dict(wind=lambda x, y: (-(y - 0.5), (x - 0.5)), wind_scale=lambda t: np.sin(2 * np.pi * t))
    args.optim = args.optim or "alltime"
V = hp.SquareMeshP1(0.0, 1.0, round(1 / dx))
n, Nt = V.nodes, round(T / dt)
tl = (Nt + 1) * n
z = lambda x0: np.concatenate([x0, np.zeros(Nt * n)])
if args.problem == "schnak":
    ic = hp.schnak_sys_IC(0, 1, dx, n, V.vertex_to_dof)
    full = hp.solve_schnak_system(np.full(tl, 0.1), z(ic[0]), z(ic[1]), V, n, Nt, dt, None, **wind_kw)   # true control a = 0.1
elif args.problem == "chtxs":
    ic = hp.chtxs_sys_IC(0, 1, dx, n, V.vertex_to_dof)
    full = hp.solve_chtxs_system(np.full(tl, 10.0), z(ic[0]), z(ic[1]), V, n, Nt, dt, None)           # c = 100 * r, r = 1/10
else:
    ic = (hp.nonlinear_equation_IC(0, 1, dx, n, V.vertex_to_dof),)
    full = (hp.solve_nonlinear_equation(np.full(tl, 0.5), z(ic[0]), None, V, n, Nt, dt, None)[0],)
opts = dict(max_iter_GD=args.iters, tol=0.0, **wind_kw)
if args.named_c3:
    opts.update(c_lower=0.0, c_upper=0.5)
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
