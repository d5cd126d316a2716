"""Embarrassingly parallel parameter sweeps over the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL on ROCm, ``gloo`` on CPU for
tests).  Independent units -- regularisation values beta (config C5), Armijo trial steps, whole PGD
problems, or their products (beta x trial) -- are dealt round-robin to the ranks; a trajectory never crosses
a GPU, so the data path has no collective.  The only exchange is one all-gather of the per-unit scalar
results (8 bytes per unit: latency bound on xGMI).
A rank's units are handed to the solver TOGETHER (``sweep_batched``): B trajectories per kernel launch are what
fills a GPU at the config sizes -- one trajectory per GPU leaves it latency bound (8 GPUs x 1 trajectory advance
fewer time steps per second than one GPU with 64).
The reference ran such sweeps as separate serial cluster jobs, one edited script copy per value
(advection_solidbody_FCT_PDECO_alltime_eddie_drift_beta0_001.py:45).
"""
from __future__ import annotations

import math


def shard(units, rank: int, world: int):
    """Indices of the units owned by ``rank`` (round-robin, deterministic)."""
    return list(range(rank, len(units), world))


def sweep(units, run_unit, dist=None, device=None):
    """Run ``run_unit(unit) -> float`` for every unit, sharded over the ranks of ``dist``
    (a ``torch.distributed`` module with an initialised process group, or None for one process).
    Returns the full list of results, in the order of ``units``, on every rank."""
    return sweep_batched(units, lambda mine: [run_unit(u) for u in mine], dist, device)


def sweep_batched(units, run_batch, dist=None, device=None):
    """Like :func:`sweep`, but a rank's whole share goes to ONE call ``run_batch(list_of_units) -> list of floats``
    (the device solvers take ``batch=len(list)``: the units advance together in every kernel launch).  Shares differ
    by at most one unit (ragged counts are padded in the all-gather, not in the work); a rank without units does not
    call ``run_batch``.  Returns the full, ordered result list on every rank."""
    def run(mine):
        if not mine:
            return []
        res = [float(x) for x in run_batch(mine)]
        if len(res) != len(mine):
            raise ValueError(f"run_batch returned {len(res)} results for {len(mine)} units")
        return res

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return run(list(units))
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard(units, rank, world)
    per_rank = math.ceil(len(units) / world)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    local = torch.full((per_rank,), float("nan"), dtype=torch.float64, device=device)
    for k, val in enumerate(run([units[idx] for idx in mine])):
        local[k] = val
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)   # the sweep's only collective
    out = [float("nan")] * len(units)
    for r in range(world):
        vals = gathered[r].cpu().tolist()
        for k, idx in enumerate(shard(units, r, world)):
            out[idx] = vals[k]
    return out
