#!/usr/bin/env python3
"""Config C5 as ONE batched line search: eight regularisation values beta x eight Armijo trial steps s0/2^k = 64 units, every
unit one forward trajectory (advection_solidbody_FCT_PDECO_alltime.py:43-74 ran one edited script copy per beta; its
line search, helpers.py:1681-1708, one trial after the other).  `sweep.sweep_batched` deals the units round-robin to the
ranks and hands a rank its whole share in ONE call, so all its trajectories advance together in every kernel launch -- what
fills a GPU at this mesh size (one GPU with 64 trajectories per launch outruns eight GPUs with one each: DESIGN.md section 6).
The only exchange is one all-gather of the 64 costs.

  python examples/c5_beta_trial_batch.py                                   # all 64 units in one batch on one GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/c5_beta_trial_batch.py   # 8 per GPU"""
import argparse
import os
import time

import numpy as np

from _common import hp, solvers, sweep, gaussian

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100, help="time steps (T = steps * dt; the config: 100)")
ap.add_argument("--betas", type=int, default=8)
ap.add_argument("--trials", type=int, default=8)
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
dist = None
if world > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

a1, a2, dx, dt = -1.0, 1.0, 0.025, 0.001
mesh = hp.SquareMeshP1(a1, a2, round((a2 - a1) / dx))
n, Nt = mesh.nodes, args.steps
tl = (Nt + 1) * n
lo, hi, s0, gam = 0.0, 5.0, 1.0, 1e-4
betas = [10.0 ** (-k / 2) for k in range(args.betas)]
svals = [s0 / 2 ** k for k in range(args.trials)]
units = [(i, k) for i in range(len(betas)) for k in range(len(svals))]
share = max(1, -(-len(units) // world))
prob = solvers.SolidBodyDrift(mesh, Nt, dt, eps=0.0, drift=(1.0, 1.0), rot_scale=0.0, device_id=local_rank, batch=share,
                              order=hp.ORDER_VERTEX)
ctx = prob.ctx
u0 = gaussian(a1, a2, dx)                                  # vertex order = device order
init = np.zeros(tl)
init[:n] = u0
uhat = ctx.array(init)
prob.forward(ctx.array(np.full(tl, 2.0)), uhat, batch=1)   # target trajectory at the true control c = 2

# the current iterate (c = 1 for every beta), its state and adjoint: the same for all beta; the descent directions are not
c = ctx.array(np.ones(tl))
u, p = ctx.array(init), ctx.zeros(tl)
prob.forward(c, u, batch=1)
prob.adjoint(c, u, uhat, p, "alltime", batch=1)
zero = ctx.zeros(tl)
misfit0 = float(prob.cost(u, uhat, c, 0.0, "alltime", batch=1)[0])
cnorm0 = float(ctx.l2_norm_sq_Q(c, zero, Nt, dt)[0])
dirs = []
for beta in betas:
    d = ctx.zeros(tl)
    prob.descent_direction(c, u, p, beta, d)
    dirs.append(d)


def run_batch(mine):
    """J(beta_i, clip(c + s_k d_i)) for the units of this rank: one batched forward sweep, one batched cost evaluation"""
    B = len(mine)
    cB, uB = ctx.zeros(B * tl), ctx.zeros(B * tl)
    uhB, cK, zB = ctx.zeros(B * tl), ctx.zeros(B * tl), ctx.zeros(B * tl)
    for b, (i, k) in enumerate(mine):
        ctx.project_control(c, svals[k], dirs[i], lo, hi, cB.ptr + 8 * b * tl, tl)
        uB.copy_from(u, n, dst_off=b * tl)                 # level 0 = the initial condition
        uhB.copy_from(uhat, tl, dst_off=b * tl)
        cK.copy_from(c, tl, dst_off=b * tl)
    ctx.synchronize()
    t0 = time.perf_counter()
    prob.forward(cB, uB, batch=B)
    ctx.synchronize()
    el = time.perf_counter() - t0
    misfit = prob.cost(uB, uhB, cB, 0.0, "alltime", batch=B)
    cn = ctx.l2_norm_sq_Q(cB, zB, Nt, dt, batch=B)
    dif = ctx.l2_norm_sq_Q(cB, cK, Nt, dt, batch=B)
    run_batch.info = (B, el, int(ctx.kernel_regime(B)))
    run_batch.dif = {u_: float(x) for u_, x in zip(mine, dif)}
    J = [float(misfit[b]) + 0.5 * betas[i] * float(cn[b]) for b, (i, k) in enumerate(mine)]
    # the combined value is cost_functional's own for that beta (helpers.py:383-441)
    i0, k0 = mine[0]
    ref = float(prob.cost(uB, uhB, cB, betas[i0], "alltime", batch=1)[0])
    assert abs(J[0] - ref) <= 1e-12 * abs(ref), (J[0], ref)
    for a in (cB, uB, uhB, cK, zB):
        a.free()
    return J


costs = sweep.sweep_batched(units, run_batch, dist)
difs = sweep.sweep_batched(units, lambda mine: [run_batch.dif[u_] for u_ in mine], dist)
B, el, regime = run_batch.info
if rank == 0:
    print(f"{len(units)} units = {len(betas)} beta x {len(svals)} trial steps on {world} rank(s): {B} trajectories per launch, "
          f"{Nt} steps in {el * 1e3:.1f} ms = {B * Nt / el:,.0f} timesteps/s per GPU (kernel regime {regime})")
    for i, beta in enumerate(betas):
        J0 = misfit0 + 0.5 * beta * cnorm0
        acc = next((k for k in range(len(svals)) if costs[i * len(svals) + k] - J0 <= -gam / svals[k] * difs[i * len(svals) + k]), None)
        print(f"beta = {beta:9.3e}   J(c) = {J0:.6e}   accepted trial: " +
              (f"k = {acc} (s = {svals[acc]:.4g}), J = {costs[i * len(svals) + acc]:.6e}" if acc is not None else "none"))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
prob.close()
