#!/usr/bin/env python3
"""Config C5: the regularisation sweep of advection_solidbody_FCT_PDECO_alltime.py (the reference ran one edited
script copy per beta as separate cluster jobs).  [-1,1]^2, 81 x 81, dt = 1e-3, T = 0.1, no rotation, Gaussian
initial condition, true control c = 2, c^0 = 1, all-time misfit; targets from the forward solve at the true control.
One process per GPU, the beta values dealt round-robin; the only exchange is an all-gather of the final costs.

  python examples/c5_beta_sweep.py                                    # all 8 values on one GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/c5_beta_sweep.py"""
import argparse
import os

import numpy as np

from _common import hp, solvers, sweep, gaussian, to_dof

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
dist = None
if world > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

a1, a2, dx, dt, T = -1.0, 1.0, 0.025, 0.001, 0.1
mesh = hp.SquareMeshP1(a1, a2, round((a2 - a1) / dx))
n, Nt = mesh.nodes, round(T / dt)
tl = (Nt + 1) * n
prob = solvers.SolidBodyDrift(mesh, Nt, dt, eps=0.0, drift=(1.0, 1.0), rot_scale=0.0, device_id=local_rank,
                              order=hp.ORDER_VERTEX)
u0 = gaussian(a1, a2, dx)                       # vertex order = device order
uhat = np.zeros(tl)
uhat[:n] = u0
uhat = prob.solve_state(2.0 * np.ones(tl), uhat)          # target trajectory at the true control c = 2
betas = [10.0 ** (-k / 2) for k in range(8)]


def run(beta):
    u, p, c, hist = solvers.pgd_solidbody_alltime(prob, u0, uhat, np.ones(tl), beta, 0.0, 5.0, args.iters)
    return hist["cost"][-1]


costs = sweep.sweep(betas, run, dist)
if int(os.environ.get("RANK", "0")) == 0:
    for b, J in zip(betas, costs):
        print(f"beta = {b:9.3e}   J after {args.iters} PGD iterations = {J:.8e}")
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
prob.close()
