"""Embarrassingly parallel parameter sweeps over the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL on ROCm, ``gloo`` on CPU for
tests).  Independent units -- regularisation values beta (config C5), Armijo trial steps, whole PGD
problems -- are dealt round-robin to the ranks; a trajectory never crosses a GPU, so the data path has
no collective.  The only exchange is one all-gather of the per-unit scalar results (8 bytes per unit:
latency bound on xGMI).
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
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(run_unit(u)) for u in units]
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard(units, rank, world)
    per_rank = math.ceil(len(units) / world)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    local = torch.full((per_rank,), float("nan"), dtype=torch.float64, device=device)
    for k, idx in enumerate(mine):
        local[k] = float(run_unit(units[idx]))
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)   # the sweep's only collective
    out = [float("nan")] * len(units)
    for r in range(world):
        vals = gathered[r].cpu().tolist()
        for k, idx in enumerate(shard(units, r, world)):
            out[idx] = vals[k]
    return out
