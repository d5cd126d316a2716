#!/usr/bin/env python3
"""Where a speculative PGD iteration of configs 3 / 4 spends its time: wraps the phases of pdeco.SystemPDECO (state batch,
adjoint, costs, control projections / copies, descent direction) with a device synchronisation + host clock.
  python3 tools/pgd_phase_times.py [schnak|chtxs]"""
import importlib
import os
import sys
import time
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hp = importlib.import_module("fem-fct-pdeco_amd")
pdeco = importlib.import_module("fem-fct-pdeco_amd.pdeco")

problem = sys.argv[1] if len(sys.argv) > 1 else "chtxs"
V = hp.SquareMeshP1(0.0, 1.0, 40)
n, Nt, dt = V.nodes, 200, 5e-4
tl = (Nt + 1) * n
v2d = V.vertex_to_dof
ic = hp.schnak_sys_IC(0, 1, 0.025, n, v2d) if problem == "schnak" else (1.5 + 0.1 * (0.5 - np.random.default_rng(5).random(n)),) * 2
ctrue, optim = (0.1, "finaltime") if problem == "schnak" else (10.0, "alltime")

P = pdeco.SystemPDECO(problem, V, Nt, dt)
c = P._up(np.full(tl, ctrue))
us = [P._up(np.concatenate([x0, np.zeros(Nt * n)])) for x0 in ic]
P._state(c, us[0], us[1], P._zeros(n), 1)
full = [P._down(x) for x in us]
P.close()
tg = [x if optim == "alltime" else x[Nt * n:] for x in full]

acc = defaultdict(float)
cnt = defaultdict(int)


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    lab = label or name

    def g(*a, **k):
        obj_ctx.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        obj_ctx.synchronize()
        acc[lab] += time.perf_counter() - t0
        cnt[lab] += 1
        return r
    setattr(obj, name, g)


with pdeco.SystemPDECO(problem, V, Nt, dt, max_iter_GD=3, tol=0.0) as P:
    obj_ctx = P.ctx
    for label in ("first run of a context (graph captures, budgets)", "second run, same context", "third run, phases synchronised"):
        if label.startswith("third"):
            for nm in ("_state", "_adjoint", "_cost", "_descent"):
                wrap(P, nm)
            wrap(P.ctx, "project_control")
            wrap(P.ctx, "l2_norm_sq_Q")
        P.ctx.synchronize()
        t0 = time.perf_counter()
        r = P.run(ic, tg, speculative=True)
        P.ctx.synchronize()
        el = time.perf_counter() - t0
        print(f"{label}: {el * 1e3:.1f} ms for the initial state + adjoint and {r['it']} iterations")
print(f"{problem}: {r['it']} iterations, {el * 1e3:.1f} ms total incl. the initial state + adjoint (with the phase syncs), trials {r['armijo_its']}")
for k in sorted(acc, key=lambda k: -acc[k]):
    print(f"  {k:18s} {cnt[k]:4d} calls  {acc[k] * 1e3:8.2f} ms  ({acc[k] / cnt[k] * 1e3:7.3f} ms each)")
print(f"  {'unaccounted':18s}             {(el - sum(acc.values())) * 1e3:8.2f} ms")
