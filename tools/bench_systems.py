#!/usr/bin/env python3
"""Throughput of the other configs' trajectory solvers (device-resident sweeps, n = 1681):
C3 Schnakenberg (dt = 5e-4), C4 chemotaxis (dt = 5e-4), nonlinear; forward + adjoint."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# crash diagnostics (FEMFCT_DUMP_MAPS=<file>): Python stack on a fatal signal and the process's address map, so that the
# raw frame addresses a native fault handler prints can be resolved against the libraries afterwards
import faulthandler
faulthandler.enable()


def dump_maps(tag):
    path = os.environ.get("FEMFCT_DUMP_MAPS")
    if path:
        with open("/proc/self/maps") as f, open(f"{path}.{tag}", "w") as g:
            g.write(f.read())

hp = importlib.import_module("fem-fct-pdeco_amd")
systems = importlib.import_module("fem-fct-pdeco_amd.systems")

V = hp.SquareMeshP1(0.0, 1.0, 40)
n = V.nodes
Nt, dt = 200, 5e-4
S = systems.PDESystems(V, order=hp.ORDER_VERTEX)
ctx = S.ctx
tl = (Nt + 1) * n
rng = np.random.default_rng(0)


def traj(u0):
    a = np.zeros(tl)
    a[:n] = u0
    return ctx.array(a)


def timeit(fn, reps=3):
    fn()
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


u0s, v0s = hp.schnak_sys_IC(0, 1, 0.025, n, np.arange(n))
c = ctx.array(0.1 + 0.01 * rng.random(n))
par, wind = systems._schnak_par()
Aw, AwT = S.convection(wind, "schnak")
u, v, p, q = traj(u0s), traj(v0s), traj(np.zeros(n)), traj(np.zeros(n))
t = timeit(lambda: ctx.schnak_forward(Aw, c, u, v, Nt, dt, par, 1.0))
kinfo = ctx.traj_krylov_info(Nt)
print(f"Schnakenberg forward : {Nt / t:8.0f} steps/s  (species-solve iters max {kinfo['solver_iters'].max()}, jacobi sweeps max {ctx.traj_info(Nt)['solver_iters'].max()})")
uh, vh = ctx.array(rng.random(n)), ctx.array(rng.random(n))
dump_maps("before_schnak_adjoint")
t = timeit(lambda: ctx.schnak_adjoint(AwT, u, v, uh, vh, p, q, Nt, dt, par))
print(f"Schnakenberg adjoint : {Nt / t:8.0f} steps/s  (species-solve iters max {ctx.traj_krylov_info(Nt)['solver_iters'].max()})")

u0c = 1.5 + 0.1 * (0.5 - rng.random(n))
u, v, p, q = traj(u0c), traj(u0c), traj(np.zeros(n)), traj(np.zeros(n))
cpar = systems._chtxs_par()
cc = ctx.array(20 * rng.random(n))
t = timeit(lambda: ctx.chtxs_forward(cc, u, v, Nt, dt, cpar, 0.1))
print(f"chemotaxis forward   : {Nt / t:8.0f} steps/s  (species-solve iters max {ctx.traj_krylov_info(Nt)['solver_iters'].max()})")
ct = ctx.array(20 * rng.random(tl))
uh, vh = ctx.array(rng.random(tl)), ctx.array(rng.random(tl))
t = timeit(lambda: ctx.chtxs_adjoint(u, v, uh, vh, p, q, ct, Nt, dt, cpar, 0.1, True))
print(f"chemotaxis adjoint   : {Nt / t:8.0f} steps/s  (species-solve iters max {ctx.traj_krylov_info(Nt)['solver_iters'].max()})")

eps, _, nwind = hp.get_nonlinear_eqns_params()
Awn, _ = S.convection(nwind, "nonlinear")
u, p = traj(hp.nonlinear_equation_IC(0, 1, 0.025, n, np.arange(n))), traj(np.zeros(n))
cn = ctx.array(rng.random(n))
t = timeit(lambda: ctx.nonlinear_forward(Awn, cn, u, Nt, 1e-3, eps))
print(f"nonlinear forward    : {Nt / t:8.0f} steps/s")
t = timeit(lambda: ctx.nonlinear_adjoint(Awn, u, uh, p, Nt, 1e-3, eps))
print(f"nonlinear adjoint    : {Nt / t:8.0f} steps/s")
S.close()

# ---------------------------------------------------------------------------------------------
# whole projected-gradient iterations of the refactored drivers (pdeco.py), configs C3 / C4 sizes:
# targets from the build's own forward solve at the true control, 3 PGD iterations each
pdeco = importlib.import_module("fem-fct-pdeco_amd.pdeco")
v2d = V.vertex_to_dof


def targets_for(problem, ic, ctrue):
    P = pdeco.SystemPDECO(problem, V, Nt, dt)
    try:
        c = P._up(np.full(tl, ctrue))
        us = [P._up(np.concatenate([x0, np.zeros(Nt * n)])) for x0 in ic]
        clev = P._zeros(n)
        P._state(c, us[0], us[1] if len(us) > 1 else None, clev, 1)
        out = [P._down(x) for x in us]
    finally:
        P.close()
    return out


for problem, ic, ctrue, optim in (
        ("schnak", hp.schnak_sys_IC(0, 1, 0.025, n, v2d), 0.1, "finaltime"),
        ("chtxs", (1.5 + 0.1 * (0.5 - np.random.default_rng(5).random(n)),) * 2, 10.0, "alltime")):
    full = targets_for(problem, ic, ctrue)
    tg = [x if optim == "alltime" else x[Nt * n:] for x in full]
    for spec in (True, False):
        with pdeco.SystemPDECO(problem, V, Nt, dt, max_iter_GD=3, tol=0.0) as P:
            P.run(ic, tg, speculative=spec)          # warm-up: graphs, budgets
        with pdeco.SystemPDECO(problem, V, Nt, dt, max_iter_GD=3, tol=0.0) as P:
            t0 = time.perf_counter()
            r = P.run(ic, tg, speculative=spec)
            el = time.perf_counter() - t0
        print(f"PGD {problem:7s} {'speculative' if spec else 'sequential '}: {(r['wall'][-1] - r['wall0']) / max(r['it'], 1) * 1e3:8.1f} ms/iteration "
              f"(whole run incl. set-up and the initial state + adjoint: {el * 1e3:.1f} ms for {r['it']} iterations) "
              f"(armijo trials {r['armijo_its']}, cost {r['cost'][0]:.4e} -> {r['cost'][-1]:.4e})")
