"""The multi-GPU sweep path on CPU: world_size 2, gloo backend (the GPU path differs only in the
backend name and the tensor device)."""
import importlib
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unit_cost(beta):
    """A cheap stand-in for 'run the PGD problem for this beta': cost functional of a fixed small
    state through the CPU oracle (tests may use the oracle)."""
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import fct as ofct
    asm = P1Assembler(SquareMesh(0, 1, 4))
    M = asm.mass()
    n = M.shape[0]
    rng = np.random.default_rng(0)
    u, t, c = rng.random(3 * n), rng.random(3 * n), rng.random(3 * n)
    return ofct.cost_functional(u, t, c, 2, 0.1, M, beta, "alltime")


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sweep = importlib.import_module("fem-fct-pdeco_amd.sweep")
    betas = [10.0 ** (-k / 2) for k in range(5)]      # odd count: ragged shards
    calls = []

    def run(b):
        calls.append(b)
        return _unit_cost(b)

    out = sweep.sweep(betas, run, dist)
    q.put((rank, out, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_sweep_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    betas = [10.0 ** (-k / 2) for k in range(5)]
    expect = [_unit_cost(b) for b in betas]
    res.sort()
    for rank, out, calls in res:
        assert out == expect                        # every rank holds the full, ordered result
        assert calls == betas[rank::2]              # each unit ran on exactly one rank
    assert sorted(res[0][2] + res[1][2]) == sorted(betas)


def _worker_batched(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sweep = importlib.import_module("fem-fct-pdeco_amd.sweep")
    # (beta x Armijo trial) units: 3 betas x 3 step sizes = 9 units over 2 ranks: shares of 5 and 4
    units = [(10.0 ** (-k / 2), 1.0 / 2 ** j) for k in range(3) for j in range(3)]
    calls = []

    def run_batch(mine):                     # ONE call per rank: its units advance as one batch
        calls.append(list(mine))
        return [_unit_cost(b) * s for b, s in mine]

    out = sweep.sweep_batched(units, run_batch, dist)
    q.put((rank, out, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_sweep_batched_world_size_2_gloo_ragged_units():
    """The partition the device solvers want: every rank gets its share of the (beta x trial) units in ONE call (one
    batch per GPU), ragged count (9 units, 2 ranks), one all-gather, full ordered result on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_batched, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    units = [(10.0 ** (-k / 2), 1.0 / 2 ** j) for k in range(3) for j in range(3)]
    expect = [_unit_cost(b) * s for b, s in units]
    res.sort()
    for rank, out, calls in res:
        assert out == expect
        assert len(calls) == 1 and calls[0] == units[rank::2]          # one batch per rank, round-robin shares
    assert len(res[0][2][0]) == 5 and len(res[1][2][0]) == 4


def test_sweep_single_process_and_shard():
    sweep = importlib.import_module("fem-fct-pdeco_amd.sweep")
    assert sweep.sweep([1.0, 2.0, 3.0], lambda b: 2 * b) == [2.0, 4.0, 6.0]
    assert sweep.shard(list(range(8)), 3, 8) == [3]
    assert sweep.shard(list(range(5)), 1, 2) == [1, 3]
    assert sweep.sweep([], lambda b: b) == []
    assert sweep.sweep_batched([1.0, 2.0, 3.0], lambda bs: [2 * b for b in bs]) == [2.0, 4.0, 6.0]
    assert sweep.sweep_batched([], lambda bs: 1 / 0) == []          # no units: the batch is not run
    with pytest.raises(ValueError):
        sweep.sweep_batched([1.0, 2.0], lambda bs: [0.0])


def test_bench_multi_rank_control_flow_under_gloo():
    """bench.py --gpus 2 with WORLD_SIZE unset: the script must start its own ranks (a child
    torch.distributed.run, nothing exec'd), rendezvous on 127.0.0.1, all-gather the per-rank cost, take the
    max-over-ranks time and print ONE JSON line from rank 0.  --stub-solver swaps the GPU problem for a stub and
    RCCL for gloo; every other line of the N > 1 path is the one the 8-GPU run executes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--stub-solver"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["stub"] is True
    # rank r's stub cost is 100 + r + beta_r with the C5 betas 10^(-r/2): both ranks' values reached rank 0
    assert r["costs_all_ranks"] == [100.0 + 0 + 1.0, 100.0 + 1 + 10.0 ** -0.5]
    # max over ranks: rank 1 sleeps 20 ms per step
    assert r["ms_per_step"] >= 20.0
    assert abs(r["value"] - 2 * 250 * 2 * 3 / (r["ms_per_step"] * 3 / 1e3)) < 1e-6 * r["value"]


def test_bench_refuses_mismatched_world_size():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-solver"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_sweep_cpu_baseline_runs_concurrent_one_core_workers():
    """bench.py --gpus N > 1 reports the sweep's CPU baseline as N concurrent 1-core oracle processes (SURVEY 8d,
    BASELINE.md 4.2): here 2 workers x (3 + 3) steps of config 5's set-up, plain CPU children."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline_sweep(2, "c5", 3)
    assert out["cores"] == 2 and out["workers_ok"] == 2 and out["kind"] == "port"
    assert out["value"] > 0 and out["unit"] == "timesteps/s"
    one = bench.cpu_worker("c2", "2", "0")
    assert one["steps"] == 4 and one["seconds"] > 0
