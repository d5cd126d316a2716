#!/usr/bin/env python3
"""Per-class kernel times of one forward sweep on the roofline mesh (tuning helper): prints one line.
usage: roof_quick.py [cells=2048] [steps=3]   (environment: the FEMFCT_* knobs under test; TAG=label)"""
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
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = 2.0 / nc
dt = 1e-3 * h / 0.025
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
for _ in range(5):
    prob.forward(d_c, d_u, batch=1)
ctx.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    prob.forward(d_c, d_u, batch=1)
ctx.synchronize()
step_ms = 1e3 * (time.perf_counter() - t0) / (steps * reps)
ctx.set_profiling(True)
prob.forward(d_c, d_u, batch=1)
rep = ctx.profile_report()
ctx.set_profiling(False)
sw = prob.solver_log(1)["solver_iters"]
u = d_u.download()
chk = float(np.abs(u[-n:]).sum())
print(f"{os.environ.get('TAG', ''):28s} N={nc + 1} step {step_ms * 1e3:8.1f} us = {1e3 / step_ms:7.1f}/s sweeps {int(sw.max()):3d} | " +
      " ".join(f"{k}={1e3 * ms / cnt:.0f}us x{cnt / steps:.1f}" for k, (ms, cnt) in rep.items() if cnt) + f" | chk {chk:.12e}", flush=True)
prob.close()
