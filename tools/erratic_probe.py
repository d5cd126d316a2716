#!/usr/bin/env python3
"""Does the one-workgroup step run at two speeds?  Per-sweep times of repeated forward sweeps on several contexts in a row.
  python3 tools/erratic_probe.py [B] [contexts] [reps]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hp = importlib.import_module("fem-fct-pdeco_amd")
solvers = importlib.import_module("fem-fct-pdeco_amd.solvers")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ncx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
nc, Nt = 40, 50
mesh = hp.SquareMeshP1(-1.0, 1.0, nc)
n = mesh.nodes
tl = (Nt + 1) * n
x, y = mesh.coordinates()
u0 = np.exp(-20 * ((x + 0.3) ** 2 + (y + 0.2) ** 2))
Bseq = [int(t) for t in os.environ.get("PROBE_BSEQ", str(B)).split(",")]
for k in range(ncx):
    B = Bseq[k % len(Bseq)]
    if os.environ.get("PROBE_ALT"):          # mesh, tiles, mesh, tiles, ... (what tools/mesh_step_check.py does)
        os.environ["FEMFCT_MESH_STEP"] = "1" if k % 2 == 0 else "0"
    prob = solvers.SolidBodyDrift(mesh, Nt, 2e-3, batch=B, order=hp.ORDER_VERTEX)
    ctx = prob.ctx
    c = ctx.array(np.tile(1.0 + 0.5 * np.sin(np.pi * x) * np.cos(np.pi * y), B * (Nt + 1)))
    init = np.zeros((B, tl)); init[:, :n] = u0
    u = ctx.array(init.ravel())
    adj = os.environ.get("PROBE_ADJ")
    if adj:
        uhat = ctx.array(np.tile(u0, B)); p = ctx.zeros(B * tl)
    ts = []
    for r in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        prob.forward(c, u, batch=B)
        if adj:
            prob.adjoint(c, u, uhat, p, "finaltime", batch=B)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / (Nt * (2 if adj else 1)) * 1e6)
    print(f"context {k}: B {B:3d} regime {ctx.kernel_regime(B)}  us/step per sweep: " + " ".join(f"{t:6.1f}" for t in ts), flush=True)
    if os.environ.get("PROBE_SLEEP"):
        time.sleep(float(os.environ["PROBE_SLEEP"]))
    prob.close()
