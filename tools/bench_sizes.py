#!/usr/bin/env python3
"""FCT steps/s of the forward sweep over the synthetic mesh sizes of SURVEY.md section 8d
(N x N nodes, rotation + drift operator at the CFL number of config C2, vertex order)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hp = importlib.import_module("fem-fct-pdeco_amd")
solvers = importlib.import_module("fem-fct-pdeco_amd.solvers")
hp.fct_helpers.VERBOSE = False

sizes = [int(a) for a in sys.argv[1:]] or [41, 81, 257, 1025, 2049, 4097]
print(f"{'N':>6s} {'nodes':>10s} {'steps':>6s} {'sweeps':>7s} {'us/step':>10s} {'steps/s':>10s} {'Mnode-steps/s':>14s}")
for N in sizes:
    nc = N - 1
    h = 2.0 / nc
    dt = 1e-3 * h / 0.025
    steps = 200 if N <= 81 else 40 if N <= 257 else 6 if N <= 1025 else 3
    mesh = hp.SquareMeshP1(-1.0, 1.0, nc)
    n = mesh.nodes
    prob = solvers.SolidBodyDrift(mesh, steps, dt, batch=1, order=hp.ORDER_VERTEX)
    ctx = prob.ctx
    x, y = mesh.coordinates()
    rng = np.random.default_rng(0)
    u0 = np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n)
    d_c = ctx.array(np.tile(1.0 + 0.5 * np.sin(3 * x) * np.cos(2 * y), steps + 1))
    init = np.zeros((steps + 1) * n)
    init[:n] = u0
    d_u = ctx.array(init)
    for _ in range(3):
        prob.forward(d_c, d_u, batch=1)
    ctx.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        prob.forward(d_c, d_u, batch=1)
    ctx.synchronize()
    t = (time.perf_counter() - t0) / (reps * steps)
    sw = int(prob.solver_log(1)["solver_iters"].max())
    print(f"{N:6d} {n:10d} {steps:6d} {sw:7d} {t * 1e6:10.1f} {1 / t:10.1f} {n / t / 1e6:14.1f}", flush=True)
    prob.close()
