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
