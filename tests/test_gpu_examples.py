"""The example drivers (examples/) keep running: one short invocation each, as a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
EX = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")


@pytest.mark.parametrize("script,args,needle", [
    ("c1_forward_solidbody.py", ["--steps", "5"], "difference of the two end states"),
    ("c2_solidbody_pdeco_finaltime.py", ["--iters", "1"], "PGD iterations in"),
    ("c3_c4_systems_pdeco.py", ["nonlinear", "--iters", "1"], "PGD iterations in"),
    ("c3_c4_systems_pdeco.py", ["schnak", "--iters", "1", "--named-c3"], "schnak (alltime): 1 PGD iterations in"),
    ("c5_beta_sweep.py", ["--iters", "1"], "beta = "),
    ("c5_beta_trial_batch.py", ["--steps", "10", "--betas", "3", "--trials", "4"], "accepted trial"),
])
def test_example_runs(script, args, needle):
    out = subprocess.run([sys.executable, os.path.join(EX, script)] + args, capture_output=True, text=True, timeout=300,
                         cwd=EX)
    assert out.returncode == 0, out.stderr[-2000:]
    assert needle in out.stdout
    if script.startswith("c1"):
        diff = float(out.stdout.split("difference of the two end states:")[1].split()[0])
        assert diff < 1e-11


def test_bench_collectives_on_rccl_with_one_rank():
    """The N > 1 leg of bench.py (RCCL init with device_id, all-gather of the cost on the GPU, barrier-bracketed fence,
    all-reduce MAX of the elapsed time, destroy) executed for real on this one-GPU box as a one-rank group."""
    import json
    root = os.path.dirname(EX)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--force-dist", "--roofline-cells", "0", "--cpu-sample", "0", "--pgd-iters", "0", "--batched="],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["value"] > 1000 and r["cost"] > 0


def test_bench_two_ranks_real_solver_on_one_gpu_gloo():
    """bench.py --gpus 2 started without a launcher: self-launch, two ranks each with its own context and regularisation
    value (both on this box's one GPU, collectives over gloo), all-gather of the two costs, max-over-ranks timing."""
    import json
    root = os.path.dirname(EX)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--backend", "gloo", "--roofline-cells", "0"], capture_output=True, text=True, timeout=400, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["parallelism"] == "beta-sweep x2"
    J = r["costs_all_ranks"]
    assert len(J) == 2 and J[0] > 0 and J[1] > 0 and J[0] != J[1]            # beta = 1 and 10^-1/2: different costs
    assert "batched" not in r and "parity" not in r                          # N = 1 legs only
    cb = r["cpu_baseline"]                                                   # the sweep's: N concurrent 1-core oracle processes
    assert cb["cores"] == 2 and cb["workers_ok"] == 2 and cb["kind"] == "port" and cb["value"] > 0
