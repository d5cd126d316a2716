#!/usr/bin/env python3
"""Launch time of the 64-patch Chebyshev kernel against the number of iterations it runs (tuning helper):
femfct_chebsi with 1..10 iterations on the roofline mesh = one k_strip4_cheb_mass launch each."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hp = importlib.import_module("fem-fct-pdeco_amd")
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = hp.Context(0)
ctx.set_mesh_square(-1.0, 1.0, nc, hp.ORDER_VERTEX)
n = ctx.n
rng = np.random.default_rng(0)
b = ctx.array(rng.random(n))
y = ctx.empty(n)
for iters in (1, 2, 3, 5, 8, 10, 19, 20):
    for _ in range(3):
        ctx.chebsi(b, y, iters)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ctx.chebsi(b, y, iters)
    ctx.synchronize()
    print(f"iters {iters:2d}: {1e6 * (time.perf_counter() - t0) / reps:8.1f} us per chebsi call", flush=True)
ctx.close()
