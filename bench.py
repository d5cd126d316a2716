#!/usr/bin/env python3
"""Benchmark of the FEM-FCT forward+adjoint hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line (rank 0).
For N > 1 it either runs under a launcher (``python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...``: RANK/LOCAL_RANK/WORLD_SIZE in the environment) or, when WORLD_SIZE is
unset, starts that launcher itself as a child process BEFORE anything touches the GPU, relays the
child's JSON line and exits with its status (the process image is never replaced).

Workload (BASELINE.json configs[1], "C2" of SURVEY.md 8d): solid-body rotation + drift control on
[-1,1]^2, 81x81 P1 nodes (dx = 0.025, n = 6561), dt = 1e-3, T = 0.25 (250 steps), eps = 0, rotation
(-y,x)*40/pi, drift b = (1,1), slotted-disc initial condition.  One bench "step" = one cost +
gradient evaluation of the projected-gradient loop: forward sweep (250 FCT steps, on-device assembly
of the drift matrices), cost functional, adjoint sweep (250 FCT steps), descent direction (251
Chebyshev solves).  Inputs (control, initial condition, target) are resident in HBM before the timed
region.  metric = FCT timesteps (forward + adjoint) per second, whole job.

N > 1: every rank runs the same sweep for its own regularisation value beta (config C5:
embarrassingly parallel) and the ranks all-gather their cost values over RCCL once per step;
"scaling": "weak".

Extra objects in the JSON line:
  roofline      HBM roofline of the dominant kernel on a large synthetic mesh (n = 2049^2: the C2
                working set is cache resident), timed live with HIP events on the library's stream.
                `achieved`/`frac` price the COMPULSORY bytes of a launch as executed (every array the
                launch touches counted once: the matrix once per multi-sweep launch) -- <= 1 by
                construction.  `one_sweep_equiv_GBps` is the figure of the textbook one-sweep
                formulation (matrix streamed once per sweep), which temporal blocking
                beats.  `traffic` = PMC-measured bytes per launch from profiles/traffic.json, used
                only when that file was produced from the same kernel sources (source_sha16).
                `one_sweep_kernels`: the same sweep with the fusions off -- the SpMV-type and
                limiter kernels of the north star, one matrix stream per launch.
  cpu_baseline  the CPU oracle (reference-faithful NumPy/SciPy restatement, SuperLU) timed on this
                host, 1 core, on a bounded sample of the same workload.
  parity        GPU trajectories of the timed run against the oracle's on the same inputs
                (relative l2, tolerance 1e-6 = BASELINE north star).
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s is achievable by a copy kernel)

# ---------------------------------------------------------------------------------------------
# Compulsory HBM bytes per matrix row of ONE LAUNCH as executed (float64 values; structured mesh in
# vertex order: index-free addressing, no column-index bytes; ELL width 7 = diagonal + 6).  Every
# array a launch reads or writes is counted once, whatever the number of sweeps it performs.
# ---------------------------------------------------------------------------------------------
def launch_bytes_per_row(fused: bool, geom_mass: bool, inline_ops: bool = False, half_d: bool = False, l_nonzero: float = 1.0,
                         rot_geom: bool = False):
    """l_nonzero: share of L's off-diagonal entries that are non-zero -- in the bandwidth regime the vanishing ones are
    neither stored by k_build_low nor loaded by k_strip4_jacobi (femfct_lowop_nonzero_fraction), so they are no part
    of the bytes a launch needs.  half_d: D stored once per edge (three slots instead of six, no diagonal)."""
    l_bytes = 8 + 6 * 8 * l_nonzero                 # diagonal + the off-diagonals that exist
    d_bytes = 3 * 8 if half_d else 7 * 8
    b = {
        "assemble": 7 * 8 * 3 + 6 * 4 + 8,          # Ad, Arot, A(write), neighbour indices, c   (per level)
        "build_low": 7 * 8 + l_bytes + d_bytes + 8 * 4,     # A, L(w), D(w), ml, u_n, b(w), x0(w)
        "jacobi": l_bytes + 8 * 3 + (1 if l_nonzero < 1.0 else 0),   # L, b, x_in, x_out(w), zero mask
        "dudt_rhs": 7 * 8 + 8 * 5,                  # A, u_L, M_diag, r(w), u_L copy(w), y1(w)
    }
    if inline_ops:                                  # operator derived in the kernels from Arot + the control's 1-ring
        b["build_low"] = 7 * 8 + l_bytes + d_bytes + 8 * 5 + 1   # Arot, L(w), D(w), c, ml, u_n, b(w), x0(w), zero mask(w)
        b["dudt_rhs"] = 7 * 8 + 8 * 6               # Arot, c, u_L, M_diag, r(w), u_L copy(w), y1(w)
        if rot_geom:                                # rotation operator evaluated from the node positions: Arot not read
            b["build_low"] -= 7 * 8
            b["dudt_rhs"] -= 7 * 8
    if fused:
        b["cheb"] = 8 * 4 + (0 if geom_mass else 7 * 8)            # b, y_mid, y_old|y_old(w), y_new(w) [+ M]
        b["flux"] = (3 * 8 if half_d else 6 * 8) + 8 * 4 + (0 if geom_mass else 6 * 8)   # D, u_L, du, ml, u_out(w) [+ M]: F never stored
    else:
        b["cheb"] = 7 * 8 + 8 * 4                                   # M, b, y_mid, y_old, y_new(w)
        b["flux"] = 6 * 8 * 3 + 8 * 5                               # M, D, F(w), u_L, du, ml, R+(w), R-(w)
        b["limit"] = 6 * 8 + 8 * 5                                  # F, R+, R-, u_L, ml, u_out(w)
    return b


# one-sweep formulation (what SURVEY.md 8d prices): bytes per row and SWEEP / iteration
ONE_SWEEP_BYTES = {"jacobi": 80, "cheb": 88, "flux": 184 + 88}


def source_sha16():
    """Identity of the kernel sources THE LOADED libfemfct.so was compiled from: the library reports the hash the Makefile
    baked into it (femfct_build_id), so a stale binary beside newer sources cannot pass for the current one.
    profiles/traffic.json carries the same value."""
    hp = importlib.import_module("fem-fct-pdeco_amd")
    return hp._lib.lib.femfct_build_id().decode()


def slotted_disc_ic(a1, a2, deltax, slit=0.05):
    """advection_solidbody_FCT_PDECO_finaltime.py:71-88 (np.arange grid, vertex order)."""
    X = np.arange(a1, a2 + deltax, deltax)
    X, Y = np.meshgrid(X, X)
    R = np.sqrt(X ** 2 + (Y - 1 / 3) ** 2)
    return ((R < 1 / 3) & ((np.abs(X) > slit) | (Y > 0.5))).astype(np.float64).reshape(-1)


def gaussian_ic(a1, a2, deltax):
    """advection_solidbody_FCT_PDECO_alltime.py:62-74 (np.arange grid, vertex order)."""
    X = np.arange(a1, a2 + deltax, deltax)
    X, Y = np.meshgrid(X, X)
    return np.exp(-20 * ((X + 2 / 3) ** 2 + 5 * (Y + 5 / 6) ** 2)).reshape(-1)


def synthetic_control(x, y, v2d, num_steps):
    """Smooth space-time control in the admissible box [0,5] (synthetic data), FEniCS DoF order."""
    t = np.linspace(0.0, 1.0, num_steps + 1)[:, None]
    c = 1.5 + 1.0 * np.sin(2 * np.pi * (x[None, :] + t)) * np.cos(np.pi * y[None, :]) + 0.5 * t
    out = np.empty_like(c)
    out[:, v2d] = c
    return np.clip(out, 0.0, 5.0).reshape(-1)


# ---------------------------------------------------------------------------------------------
# multi-rank plumbing (kept in small functions so that the gloo CPU test drives the same code)
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(argv, n_ranks):
    """Start ``python -m torch.distributed.run`` with n_ranks ranks of this script as a CHILD process (nothing
    in this process has touched the GPU yet), relay its output, return its exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


class Ranks:
    """The three things the timed region needs from torch.distributed: the all-gather of the cost
    scalars, the barrier-bracketed fence and the max-over-ranks of the elapsed time."""

    def __init__(self, world, rank, local_rank, backend, force=False):
        self.world, self.rank, self.local_rank, self.backend = world, rank, local_rank, backend
        self.dist = None
        self.torch = None
        if world > 1 or backend == "nccl":
            import torch
            self.torch = torch
        if world > 1 or force:
            if world == 1:          # --force-dist without a launcher: a one-rank group on the loopback
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            import torch.distributed as dist
            self.dist = dist
            if backend == "nccl":                                   # RCCL on ROCm
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo")
        self.device = f"cuda:{local_rank}" if backend == "nccl" else "cpu"

    def gather_costs(self, J):
        """the sweep's only exchange: one all-gather of 8 bytes per rank"""
        if self.dist is None:
            return [float(J)]
        t = self.torch.tensor([float(J)], dtype=self.torch.float64, device=self.device)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def fence(self, sync_device):
        sync_device()
        if self.backend == "nccl":
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            if self.backend == "nccl":
                self.torch.cuda.synchronize()

    def max_elapsed(self, elapsed):
        if self.dist is None:
            return elapsed
        t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


class StubProblem:
    """--stub-solver: stands in for the GPU problem so that the N > 1 control flow (self-launch, rendezvous,
    all-gather, barriers, max-over-ranks, rank-0 JSON) runs on a CPU box under gloo.  Its numbers mean nothing."""

    def __init__(self, rank):
        self.rank = rank

    def one_step(self, beta):
        time.sleep(0.01 * (1 + self.rank))
        return 100.0 + self.rank + beta

    def synchronize(self):
        pass



# ---------------------------------------------------------------------------------------------
# configs 3 and 4 (BASELINE.json configs[2], configs[3]): Schnakenberg and chemotaxis systems on the
# UnitSquare 41 x 41 mesh, dt = 5e-4, 200 forward + 200 adjoint FCT steps, device-resident sweeps
#   helpers.py:511-698 (solve_schnak_system / solve_adjoint_schnak_system), :1250-1581 (chemotaxis)
# ---------------------------------------------------------------------------------------------
def bench_systems(hp, batches=(1, 20), oracle=True, reps=3, pgd=True):
    """Per system: forward + adjoint timesteps/s with everything resident in HBM; `parity` of that very run against the
    CPU oracle on the same inputs (relative l2, tolerance 1e-6); the oracle's own 1-core rate on those inputs as
    `cpu_baseline`; the same sweeps with B trajectories per launch (the Armijo trials of helpers.py:1583-1713)."""
    systems = importlib.import_module("fem-fct-pdeco_amd.systems")
    V = hp.SquareMeshP1(0.0, 1.0, 40)
    n, Nt, dt = V.nodes, 200, 5e-4
    tl = (Nt + 1) * n
    S = systems._system(V)
    ctx, v2d = S.ctx, S.v2d
    rng = np.random.default_rng(31)
    out = {}

    def to_dev(x):          # FEniCS DoF order (what the reference's arrays are in) -> the device's vertex order
        return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, n)[:, v2d]).ravel()

    def from_dev(x):
        o = np.empty((x.size // n, n))
        o[:, v2d] = x.reshape(-1, n)
        return o.ravel()

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

    def traj0(x0, B=1):
        a = np.zeros((B, tl))
        a[:, :n] = to_dev(x0)
        return ctx.array(a.ravel())

    def timed(fn):
        # best of `reps` sweeps: a sweep that follows a large device-to-host download is occasionally stalled by 40-80 ms on
        # this platform; the sweeps themselves repeat to 0.1 %
        for _ in range(2):
            fn()
        best = float("inf")
        for _ in range(reps):
            ctx.synchronize()
            t0 = time.perf_counter()
            fn()
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    par_s, wind = systems._schnak_par()
    Aw, AwT = S.convection(wind, "schnak")
    par_c = systems._chtxs_par()
    u0s, v0s = hp.schnak_sys_IC(0, 1, 0.025, n, V.vertex_to_dof)
    u0c, v0c = hp.chtxs_sys_IC(0, 1, 0.025, n, V.vertex_to_dof)
    ctrl_s = 0.1 + 0.05 * rng.random(tl)           # (the forward solvers use level 1 of it for every step: SURVEY 8a quirk 1)
    ctrl_c = 20 * rng.random(tl)
    for name in ("schnakenberg", "chemotaxis"):
        sch = name == "schnakenberg"
        u0, v0, ctrl = (u0s, v0s, ctrl_s) if sch else (u0c, v0c, ctrl_c)
        entry = {"config": ("C3 Schnak_FCT_PDECO (helpers.py:511-698): UnitSquare 41x41 P1 (n=1681), dt=5e-4, 200 fwd + 200 adj steps, "
                            "final-time adjoint" if sch else
                            "C4 chemotaxis_FCT_PDECO_AT (helpers.py:1250-1581): UnitSquare 41x41 P1 (n=1681), dt=5e-4, 200 fwd + 200 adj "
                            "steps, all-time adjoint, rescaling 1/10"),
                 "unit": "timesteps/s", "batched": []}
        for B in batches:
            u, v, p, q = traj0(u0, B), traj0(v0, B), ctx.zeros(B * tl), ctx.zeros(B * tl)
            c1 = ctx.array(np.tile(to_dev(ctrl[n:2 * n]), B))
            if sch:
                uhT, vhT = ctx.array(np.tile(to_dev(0.9 * u0), B)), ctx.array(np.tile(to_dev(1.1 * v0), B))
                fwd = lambda: ctx.schnak_forward(Aw, c1, u, v, Nt, dt, par_s, 1.0, batch=B)
                adj = lambda: ctx.schnak_adjoint(AwT, u, v, uhT, vhT, p, q, Nt, dt, par_s, batch=B)
            else:
                call = ctx.array(np.tile(to_dev(ctrl), B))
                uh, vh = ctx.array(np.tile(to_dev(np.tile(0.9 * u0 + 0.01, Nt + 1)), B)), ctx.array(np.tile(to_dev(np.tile(1.05 * v0, Nt + 1)), B))
                fwd = lambda: ctx.chtxs_forward(c1, u, v, Nt, dt, par_c, 0.1, batch=B)
                adj = lambda: ctx.chtxs_adjoint(u, v, uh, vh, p, q, call, Nt, dt, par_c, 0.1, True, batch=B)
            tf = timed(fwd)
            ta = timed(adj)
            val = 2 * Nt * B / (tf + ta)
            if B == 1:
                kin = ctx.traj_krylov_info(Nt)
                entry.update({"value": val, "forward_steps_per_s": Nt / tf, "adjoint_steps_per_s": Nt / ta,
                              "kernel_regime": int(ctx.kernel_regime(1)), "graph_replay": bool(ctx.graph_replay_active()),
                              "species_solve_iters_max": int(kin["solver_iters"].max())})
                if oracle:
                    from oracle.mesh import SquareMesh
                    from oracle.assembly import P1Assembler
                    from oracle import traj as otraj
                    asm = P1Assembler(SquareMesh(0.0, 1.0, 40))
                    ug, vg = from_dev(u.download()), from_dev(v.download())
                    pg, qg = from_dev(p.download()), from_dev(q.download())
                    uo, vo = np.zeros(tl), np.zeros(tl)
                    uo[:n], vo[:n] = u0, v0
                    z = lambda: np.zeros(tl)
                    t0 = time.perf_counter()
                    if sch:
                        otraj.solve_schnak_system(ctrl, uo, vo, asm, n, Nt, dt)
                        po, qo = otraj.solve_adjoint_schnak_system(uo, vo, 0.9 * u0, 1.1 * v0, z(), z(), Nt * dt, asm, n, Nt, dt)
                    else:
                        otraj.solve_chtxs_system(ctrl, uo, vo, asm, n, Nt, dt)
                        po, qo = otraj.solve_adjoint_chtxs_system(uo, vo, np.tile(0.9 * u0 + 0.01, Nt + 1), np.tile(1.05 * v0, Nt + 1),
                                                                  z(), z(), ctrl, Nt * dt, asm, n, Nt, dt, None, "alltime")
                    t_cpu = time.perf_counter() - t0
                    errs = {"u_rel_l2": rel(ug, uo), "v_rel_l2": rel(vg, vo), "p_rel_l2": rel(pg, po), "q_rel_l2": rel(qg, qo)}
                    entry["parity"] = dict(errs, tolerance=1e-6, ok=bool(max(errs.values()) < 1e-6),
                                           against="oracle/traj.py on the inputs of the timed run (adjoint: of the GPU's own states vs the oracle's own)")
                    entry["cpu_baseline"] = {"value": 2 * Nt / t_cpu, "unit": "timesteps/s", "cores": 1, "kind": "port",
                                             "sample": f"the same {Nt} forward + {Nt} adjoint steps, oracle (NumPy/SciPy, SuperLU), 1 process",
                                             "host_cpus": os.cpu_count()}
            entry["batched"].append({"batch_per_gpu": B, "value": val, "unit": "timesteps/s", "kernel_regime": int(ctx.kernel_regime(B))})
            for a in ([u, v, p, q, c1] + ([uhT, vhT] if sch else [call, uh, vh])):
                a.free()
        out[name] = entry
    if pgd:
        # whole PGD iterations of the refactored drivers (Schnak_FCT_PDECO_refactored.py:160-262,
        # chemotaxis_FCT_PDECO_AT_refactored.py:160-270): all Armijo trials of an iteration as one batch vs one by one
        pdeco = importlib.import_module("fem-fct-pdeco_amd.pdeco")
        for name, problem, ic, ctrue in (("schnakenberg", "schnak", (u0s, v0s), 0.1), ("chemotaxis", "chtxs", (u0c, v0c), 10.0)):
            optim = pdeco.DEFAULTS[problem]["optim"]
            with pdeco.SystemPDECO(problem, V, Nt, dt) as P:      # targets: the build's own forward solve at a constant control
                c = P._up(np.full(tl, ctrue))
                us = [P._up(np.concatenate([x0, np.zeros(Nt * n)])) for x0 in ic]
                P._state(c, us[0], us[1], P._zeros(n), 1)
                full = [P._down(x) for x in us]
            tg = [x if optim == "alltime" else x[Nt * n:] for x in full]
            rec = {}
            for spec in (True, False):
                with pdeco.SystemPDECO(problem, V, Nt, dt, max_iter_GD=3, tol=0.0) as P:
                    P.run(ic, tg, speculative=spec)               # first run: graph captures, sweep budgets settle
                    P.ctx.synchronize()
                    t0 = time.perf_counter()
                    r = P.run(ic, tg, speculative=spec)
                    el = time.perf_counter() - t0
                per_it = sorted(b - a for a, b in zip([r["wall0"]] + r["wall"][:-1], r["wall"]))
                # speculative: every iteration evaluates all trials, so the median iteration is the typical one (an occasional
                # stalled sweep stays out); sequential: the iterations differ by their trial counts -> mean
                rec["speculative" if spec else "sequential"] = {"ms_per_pgd_iteration": 1e3 * (per_it[len(per_it) // 2] if spec else sum(per_it) / len(per_it)),
                                                                "ms_per_pgd_iteration_mean": 1e3 * (r["wall"][-1] - r["wall0"]) / max(r["it"], 1),
                                                                "ms_whole_run_incl_setup_and_initial_solves": 1e3 * el,
                                                                "armijo_trials": [int(k) for k in r["armijo_its"]],
                                                                "cost": [float(r["cost"][0]), float(r["cost"][-1])]}
            rec["max_iter_armijo"] = int(pdeco.DEFAULTS[problem]["max_iter_armijo"])
            out[name]["pgd"] = rec
    return out


def bench_small_mesh_batches(hp, solvers, device_id, batches=(1, 8, 64, 256), Nt=50):
    """The 41 x 41 mesh of configs 3 / 4 with the solid-body operator, B trajectories per launch: the one-workgroup-per-
    trajectory step (kernels_mesh.hip, the default for N <= 42) against the tile path (FEMFCT_MESH_STEP=0) on the same
    inputs -- forward + final-time adjoint timesteps/s and the largest relative l2 difference between the two paths."""
    nc = 40
    mesh = hp.SquareMeshP1(-1.0, 1.0, nc)
    n, dt = mesh.nodes, 1e-3 * 80 / nc
    tl = (Nt + 1) * n
    xs = np.linspace(-1, 1, nc + 1)
    X, Y = np.meshgrid(xs, xs)
    u0 = (np.exp(-20 * ((X + 0.3) ** 2 + (Y + 0.2) ** 2)) + ((X - 0.3) ** 2 + (Y - 0.3) ** 2 < 0.09)).reshape(-1)
    rows, ref = [], {}
    prev = os.environ.get("FEMFCT_MESH_STEP")
    try:
        for path in ("mesh", "tiles"):
            os.environ["FEMFCT_MESH_STEP"] = "1" if path == "mesh" else "0"
            for B in batches:
                prob = solvers.SolidBodyDrift(mesh, Nt, dt, batch=B, device_id=device_id, order=hp.ORDER_VERTEX)
                c = prob.ctx
                try:
                    r2 = np.random.default_rng(1)
                    amp = 0.5 + r2.random((B, 1, 1))
                    cks = (amp * (1.0 + 0.5 * np.sin(np.pi * X.reshape(1, 1, n)) * np.cos(np.pi * Y.reshape(1, 1, n))) * np.ones((B, Nt + 1, 1))).reshape(-1)
                    init = np.zeros((B, tl)); init[:, :n] = u0
                    dc, du, dp = c.array(cks), c.array(init.reshape(-1)), c.zeros(B * tl)
                    duh = c.array(np.tile(u0, B))

                    def sweep():
                        prob.forward(dc, du, batch=B)
                        prob.adjoint(dc, du, duh, dp, "finaltime", batch=B)

                    for _ in range(4):          # (budgets and graphs settle)
                        sweep()
                    el = float("inf")
                    for _ in range(4):          # best of four: this loop downloads whole batches between its timings (see bench_systems.timed)
                        c.synchronize()
                        t0 = time.perf_counter()
                        sweep()
                        c.synchronize()
                        el = min(el, time.perf_counter() - t0)
                    u, p = du.download(), dp.download()
                    row = {"path": path, "batch_per_gpu": B, "value": 2 * Nt * B / el, "unit": "timesteps/s", "us_per_step": 1e6 * el / (2 * Nt),
                           "kernel_regime": int(c.kernel_regime(B)), "sweeps_max": int(prob.solver_log(B)["solver_iters"].max())}
                    if path == "mesh":
                        ref[B] = (u, p)
                    else:
                        row["rel_l2_vs_mesh_path"] = max(float(np.linalg.norm(u - ref[B][0]) / np.linalg.norm(u)),
                                                         float(np.linalg.norm(p - ref[B][1]) / np.linalg.norm(p)))
                    rows.append(row)
                finally:
                    prob.close()
    finally:
        if prev is None:
            os.environ.pop("FEMFCT_MESH_STEP", None)
        else:
            os.environ["FEMFCT_MESH_STEP"] = prev
    return {"config": f"solid-body operator on the 41x41 mesh of configs 3/4 (n=1681), {Nt} fwd + {Nt} adj steps, smooth controls", "rows": rows}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="independent trajectories per launch on each GPU")
    ap.add_argument("--roofline-cells", type=int, default=2048, help="cells per side of the roofline mesh (0: skip)")
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=250,
                    help="forward+adjoint oracle steps each; 250 = the whole C2 sweep, ~10 s of one core (0: skip)")
    ap.add_argument("--pgd-iters", type=int, default=5, help="projected-gradient iterations of the C2 problem (0: skip)")
    ap.add_argument("--batched", type=str, default="8,64,256", help="extra batch sizes reported in 'batched' ('' : skip)")
    ap.add_argument("--tolerance-table", type=int, default=1, help="1: low-order solve tolerance 1e-13 / 1e-11 / 1e-9: sweeps, large-mesh steps/s, C2 parity")
    ap.add_argument("--systems", type=int, default=1, help="1: add configs 3 and 4 (Schnakenberg, chemotaxis; 41x41, 200 + 200 steps) as 'systems'")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (rehearses the RCCL "
                         "calls of the N > 1 path on a one-GPU box)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo + fewer GPUs than ranks (ranks share GPUs round-robin) rehearses the whole "
                         "N > 1 run, real solver included, on a one-GPU box -- not a scaling measurement")
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 (default, the configuration BASELINE.json quotes the metric on): "
                         "advection_solidbody_FCT_PDECO_finaltime, 250 + 250 steps, final-time misfit.  c5: the set-up of "
                         "advection_solidbody_FCT_PDECO_alltime.py:43-74 (config 5's sweep member): 100 + 100 steps, no "
                         "rotation, Gaussian initial condition, all-time misfit, target = own forward solve at c = 2")
    ap.add_argument("--cpu-worker", type=str, default="", help=argparse.SUPPRESS)   # internal: one 1-core oracle process
    ap.add_argument("--stub-solver", action="store_true",
                    help="CPU rehearsal of the multi-rank control flow (gloo, no GPU, no library); not a measurement")
    args = ap.parse_args()

    if args.cpu_worker:                      # child of cpu_baseline_sweep: CPU only, never touches the GPU or the library
        print(json.dumps(cpu_worker(*args.cpu_worker.split(","))), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    # The result stream carries ONE line.  Native libraries write to file descriptor 1 as well (RCCL prints a version banner
    # on its first communicator): keep the real stdout aside and point fd 1 at stderr for the rest of the run.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(res):
        os.write(result_fd, (json.dumps(res) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ranks = Ranks(world, rank, local_rank, "gloo" if args.stub_solver else args.backend, force=args.force_dist)
    device_id = local_rank
    if args.backend == "gloo" and not args.stub_solver:
        import torch
        device_id = local_rank % max(1, torch.cuda.device_count())     # rehearsal: ranks may share a GPU

    betas = [10.0 ** (-k / 2) for k in range(8)]          # C5 sweep values
    beta = betas[rank % len(betas)] if world > 1 else 1.0  # C2: beta = 1
    c5 = args.workload == "c5"
    a1, a2, deltax, dt, T = -1.0, 1.0, 0.1 / 2 / 2, 0.001, (0.1 if c5 else 0.25)
    n_cells = round((a2 - a1) / deltax)
    Nt = round(T / dt)
    om = np.pi / 40
    rot_scale = 0.0 if c5 else 1.0           # alltime.py:146 multiplies Arot by 0
    optim = "alltime" if c5 else "finaltime"
    if c5 and world == 1:
        beta = betas[3]                       # 10^-1.5, the value of tests/test_gpu_fullsize.py::test_c5_alltime_sweep_setup
    B = args.batch

    if args.stub_solver:
        stub = StubProblem(rank)
        one_step = lambda: ranks.gather_costs(stub.one_step(beta))
        sync = stub.synchronize
        n = (n_cells + 1) ** 2
    else:
        hp = importlib.import_module("fem-fct-pdeco_amd")
        solvers = importlib.import_module("fem-fct-pdeco_amd.solvers")
        hp.fct_helpers.VERBOSE = False
        mesh = hp.SquareMeshP1(a1, a2, n_cells)
        n = mesh.nodes
        tl = (Nt + 1) * n
        # device arrays live in dolfin vertex order (the library's fast layout: index-free stencil
        # addressing + 2-D tile kernels); the DoF-ordered arrays below feed the CPU oracle
        prob = solvers.SolidBodyDrift(mesh, Nt, dt, om=om, eps=0.0, rot_scale=rot_scale, batch=B, device_id=device_id,
                                      order=hp.ORDER_VERTEX)
        ctx = prob.ctx
        to_dev = lambda x: hp.reorder_vector_from_dof(x, x.size // n, n, mesh.vertex_to_dof)
        from_dev = lambda x: hp.reorder_vector_to_dof(x, x.size // n, n, mesh.vertex_to_dof)
        xs, ys = mesh.coordinates()
        if c5:
            u0 = hp.reorder_vector_to_dof(gaussian_ic(a1, a2, deltax), 1, n, mesh.vertex_to_dof)
            ck = np.ones(tl)                                             # c^0 = 1 (alltime.py:164)
        else:
            u0 = hp.reorder_vector_to_dof(slotted_disc_ic(a1, a2, deltax), 1, n, mesh.vertex_to_dof)
            ck = synthetic_control(xs, ys, mesh.vertex_to_dof, Nt)
        init = np.zeros((B, tl))
        init[:, :n] = to_dev(u0)
        d_c = ctx.array(np.tile(to_dev(ck), B))
        d_u = ctx.array(init.reshape(-1))
        if c5:      # target trajectory = the library's own forward solve at the true control c = 2 (alltime.py:93-123)
            d_c2 = ctx.array(np.full(tl, 2.0))
            d_t = ctx.array(init[0])
            prob.forward(d_c2, d_t, batch=1)
            uhat = from_dev(d_t.download())
            d_c2.free(); d_t.free()
        else:
            gpath = os.path.join(ROOT, "tests", "golden", "solidbody_t0.25_u.npz")
            uhat = np.load(gpath)["u"] if os.path.exists(gpath) else np.roll(u0, 7)
        d_p = ctx.zeros(B * tl)
        d_d = ctx.zeros(tl)
        d_rhs = ctx.empty(tl)
        d_uhat = ctx.array(np.tile(to_dev(uhat), B))
        sync = ctx.synchronize

        def one_step():
            prob.forward(d_c, d_u, batch=B)
            J = prob.cost(d_u, d_uhat, d_c, beta, optim, batch=B)
            prob.adjoint(d_c, d_u, d_uhat, d_p, optim, batch=B)
            prob.descent_direction(d_c, d_u, d_p, beta, d_d, scratch=d_rhs)   # batch member 0 (one control)
            return ranks.gather_costs(J[0])

    for _ in range(args.warmup):
        one_step()
    ranks.fence(sync)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Js = one_step()
    ranks.fence(sync)
    elapsed = ranks.max_elapsed(time.perf_counter() - t0)
    value = 2 * Nt * B * world * args.steps / elapsed

    result = {
        "metric": "FCT timesteps/sec (fwd+adj)", "value": value, "unit": "timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("C5 advection_solidbody_FCT_PDECO_alltime set-up: [-1,1]^2 81x81 P1 (dx=0.025, n=6561), dt=1e-3, "
                                "no rotation, Gaussian IC, all-time misfit, 100 fwd + 100 adj FCT steps per cost+gradient evaluation"
                                if c5 else
                                "C2 advection_solidbody_FCT_PDECO_finaltime: [-1,1]^2 81x81 P1 (dx=0.025, n=6561), "
                                "dt=1e-3, 250 fwd + 250 adj FCT steps per cost+gradient evaluation"),
                   "beta": beta,
                   "nodes": n, "num_steps": Nt, "dt": dt, "batch_per_gpu": B,
                   "parallelism": f"beta-sweep x{world}" if world > 1 else "single trajectory"},
        "cost": Js[0], "costs_all_ranks": Js if world > 1 else None,
    }
    if args.stub_solver:
        result["stub"] = True
        result["data"] = "stub (control-flow rehearsal, not a measurement)"
        ranks.close()
        if rank == 0:
            emit(result)
        return

    log = prob.solver_log(B)
    result["config"].update({"low_order_solver": "jacobi", "jacobi_sweeps_max": int(log["solver_iters"].max()),
                             "solver_resid_max": float(log["solver_resid"].max()), "graphs": True,
                             "graph_replay": bool(ctx.graph_replay_active()),   # False under rocprofv3
                             "kernel_regime": int(ctx.kernel_regime(B)), "source_sha16": source_sha16()})

    # ---------------------------------------------------- per-kernel timing at C2 size
    if rank == 0:
        ctx.set_profiling(True)
        prob.forward(d_c, d_u, batch=B)
        rep = ctx.profile_report()
        ctx.set_profiling(False)
        result["kernels_c2"] = {k: {"avg_us": 1e3 * ms / cnt, "launches_per_sweep": cnt}
                                for k, (ms, cnt) in rep.items() if cnt}

    # ------------------------------------------------------------ roofline mesh
    if rank == 0 and args.roofline_cells > 0:
        result["roofline"] = roofline(hp, solvers, args.roofline_cells, args.roofline_steps, device_id)
    if rank == 0 and world == 1 and args.batched:
        # the same sweep with B independent trajectories per launch (beta values / Armijo trials on one GPU)
        result["batched"] = []
        for Bx in [int(t) for t in args.batched.split(",") if t]:
            cb = ctx.array(np.tile(to_dev(ck), Bx))
            ib = np.zeros((Bx, tl))
            ib[:, :n] = to_dev(u0)
            ub, pb = ctx.array(ib.reshape(-1)), ctx.zeros(Bx * tl)
            uhb = ctx.array(np.tile(to_dev(uhat), Bx))

            def sweep():
                prob.forward(cb, ub, batch=Bx)
                prob.cost(ub, uhb, cb, beta, optim, batch=Bx)
                prob.adjoint(cb, ub, uhb, pb, optim, batch=Bx)

            for _ in range(2):
                sweep()
            el = float("inf")
            for _ in range(3):                 # best of three (see bench_systems.timed)
                ctx.synchronize()
                t0 = time.perf_counter()
                sweep()
                ctx.synchronize()
                el = min(el, time.perf_counter() - t0)
            # which kernels a batch of Bx small trajectories runs: per-class launch time (HIP events) and the compulsory
            # bytes of a launch over all Bx members / that time
            regime_b = int(ctx.kernel_regime(Bx))
            ctx.set_profiling(True)
            prob.forward(cb, ub, batch=Bx)
            repb = ctx.profile_report()
            ctx.set_profiling(False)
            infob = ctx.launch_info() if regime_b == 3 else None
            bprb = launch_bytes_per_row(fused=regime_b >= 2 and repb["limit"][1] == 0, geom_mass=os.environ.get("FEMFCT_GEOM_MASS", "1") != "0",
                                        inline_ops=repb["assemble"][1] == 0, half_d=regime_b == 3 and os.environ.get("FEMFCT_HALF_D", "1") != "0",
                                        l_nonzero=ctx.lowop_nonzero_fraction() if regime_b == 3 else 1.0,
                                        rot_geom=ctx.rotation_derived())
            ktab = {}
            for k, (ms, cnt) in repb.items():
                if cnt:
                    e = {"launches_per_step": cnt / Nt, "avg_launch_us": 1e3 * ms / cnt}
                    if k in bprb and k != "assemble":
                        e["compulsory_GBps"] = bprb[k] * n * Bx / (1e6 * ms / cnt)
                        e["frac"] = e["compulsory_GBps"] / HBM_PEAK_GBS
                    ktab[k] = e
            result["batched"].append({"batch_per_gpu": Bx, "value": 2 * Nt * Bx / el, "unit": "timesteps/s",
                                      "ms_per_step": 1e3 * el, "kernel_regime": regime_b,
                                      "regime_name": {0: "one-sweep row kernels", 1: "row strips", 2: "32-patch tiles (latency regime)",
                                                      3: "64-patch register strips (bandwidth regime)",
                                                     4: "one workgroup per trajectory (whole step in one launch)"}.get(regime_b, "?"),
                                      "launch_info": infob, "kernels": ktab})
            for a in (cb, ub, pb, uhb):
                a.free()
    if rank == 0 and world == 1 and args.systems:
        result["systems"] = bench_systems(hp, oracle=args.cpu_sample > 0)
        result["small_mesh_batches"] = bench_small_mesh_batches(hp, solvers, device_id)
    if rank == 0 and world == 1 and args.pgd_iters > 0 and not c5:
        # the full optimisation loop of configs[1] (finaltime_Garvie.py:164-330), everything in HBM;
        # speculative = all 10 Armijo trial steps as one batch of independent trajectories
        pg = {}
        for spec in (True, False):
            # one untimed iteration first: the graphs of this batch size are captured and its sweep budgets settle
            solvers.pgd_solidbody_finaltime(prob, to_dev(u0), to_dev(uhat), np.ones(tl), 1.0, 0.0, 5.0, 1, speculative=spec)
            prob.ctx.synchronize()
            t0 = time.perf_counter()
            _, _, _, hist = solvers.pgd_solidbody_finaltime(prob, to_dev(u0), to_dev(uhat), np.ones(tl), 1.0, 0.0, 5.0,
                                                            args.pgd_iters, speculative=spec)
            # median iteration (each ends with a cost read-back, hist["wall"]): the loop's set-up and an occasional stalled
            # sweep stay out of it; the mean over the whole call is reported beside it
            walls = [hist["wall0"]] + hist["wall"]
            per_it = sorted(b - a for a, b in zip(walls[:-1], walls[1:]))
            dt_it = per_it[len(per_it) // 2]
            pg["speculative" if spec else "sequential"] = {
                "s_per_pgd_iteration": dt_it, "s_per_pgd_iteration_mean_incl_setup": (time.perf_counter() - t0) / len(hist["cost"]),
                "armijo_trials": hist["armijo_k"], "cost": hist["cost"][-1],
                "armijo_margin_min": hist["armijo_margin_min"]}
        pg["cost_rel_diff"] = abs(pg["speculative"]["cost"] - pg["sequential"]["cost"]) / abs(pg["sequential"]["cost"])
        pg["note"] = ("every line search exhausts on this data (slotted disc, reference target of another code version): the "
                      "reference's behaviour, see DESIGN.md section 5; armijo_margin_min = smallest |J_trial - J_k + gam/s ||dc||^2| / |J_k| "
                      "over the trials looked at")
        result["pgd_c2"] = pg
        # a loop whose decisions are mixed (config 5's set-up at beta = 1e-3: the first search exhausts, the next ones accept
        # at the first trial; tests/test_gpu_fullsize.py compares it with the oracle loop)
        from types import SimpleNamespace
        Nt5 = 100
        p5 = solvers.SolidBodyDrift(mesh, Nt5, dt, om=om, eps=0.0, rot_scale=0.0, batch=1, device_id=device_id, order=hp.ORDER_VERTEX)
        try:
            tl5 = (Nt5 + 1) * n
            g0 = gaussian_ic(a1, a2, deltax)                       # vertex order = device order
            c2 = p5.ctx.array(np.full(tl5, 2.0))
            tgt = p5.ctx.array(np.concatenate([g0, np.zeros(tl5 - n)]))
            p5.forward(c2, tgt, batch=1)
            uhat5 = tgt.download()
            c2.free(); tgt.free()
            mix = {}
            for spec in (True, False):
                t0 = time.perf_counter()
                _, _, _, h5 = solvers.pgd_solidbody_alltime(p5, g0, uhat5, np.ones(tl5), 1e-3, 0.0, 5.0, 3, max_armijo=6, speculative=spec)
                mix["speculative" if spec else "sequential"] = {
                    "s_per_pgd_iteration": sorted(b - a for a, b in zip([h5["wall0"]] + h5["wall"][:-1], h5["wall"]))[len(h5["wall"]) // 2],
                    "s_per_pgd_iteration_mean_incl_setup": (time.perf_counter() - t0) / len(h5["cost"]), "armijo_trials": h5["armijo_k"],
                    "cost": h5["cost"][-1], "armijo_margin_min": h5["armijo_margin_min"]}
            mix["same_decisions"] = mix["speculative"]["armijo_trials"] == mix["sequential"]["armijo_trials"]
            mix["cost_rel_diff"] = abs(mix["speculative"]["cost"] - mix["sequential"]["cost"]) / abs(mix["sequential"]["cost"])
            result["pgd_c5_mixed_decisions"] = mix
        finally:
            p5.close()
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        # what the timed region left in HBM: batch member 0's state and adjoint trajectories for the control ck
        gpu_u = from_dev(d_u.download()[:tl])
        gpu_p = from_dev(d_p.download()[:tl])
        base, par, (uk_o, pk_o, ns_o) = cpu_baseline(a1, a2, n_cells, Nt, dt, om, u0, ck, uhat, args.cpu_sample, gpu_u, gpu_p,
                                                     rot_scale=rot_scale, optim=optim)
        result["cpu_baseline"] = base
        result["parity"] = par
        if args.tolerance_table and args.roofline_cells > 0 and ns_o == Nt and B == 1:
            relf = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

            def c2_eval(tol):
                ctx.set_solver(rel_tol=tol)
                for _ in range(2):
                    one_step()
                ctx.synchronize()
                t0 = time.perf_counter()
                one_step()
                ctx.synchronize()
                el = time.perf_counter() - t0
                lg = prob.solver_log(B)
                return {"timesteps_per_s": 2 * Nt / el, "sweeps_max": int(lg["solver_iters"].max()),
                        "u_rel_l2": relf(from_dev(d_u.download()[:tl]), uk_o), "p_rel_l2": relf(from_dev(d_p.download()[:tl]), pk_o)}

            result["tolerance_table"] = tolerance_table(hp, solvers, device_id, args.roofline_cells, args.roofline_steps, c2_eval)
            ctx.set_solver(rel_tol=1e-13)
    ranks.close()                 # (the ranks' collectives are over: what follows is rank 0's CPU-only leg)
    if rank == 0 and world > 1 and args.cpu_sample > 0 and not args.stub_solver:
        # the sweep's CPU baseline: N concurrent 1-core oracle processes, one per sweep
        # member, on this node's host cores -- plain CPU children (they import NumPy / SciPy and oracle/ only)
        result["cpu_baseline"] = cpu_baseline_sweep(world, args.workload, min(args.cpu_sample, Nt))
    if rank == 0:
        emit(result)


def cpu_worker(workload, sample, beta_index):
    """One member of the sweep's CPU baseline: forward + adjoint oracle sweep of `sample` steps on one thread."""
    os.environ["OMP_NUM_THREADS"] = "1"
    sample, beta_index = int(sample), int(beta_index)
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import traj as otraj
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:  # pragma: no cover
        pass
    a1, a2, deltax, dt = -1.0, 1.0, 0.025, 1e-3
    nc = round((a2 - a1) / deltax)
    mesh = SquareMesh(a1, a2, nc)
    asm = P1Assembler(mesh)
    n = mesh.nodes
    v2d = mesh.vertex_to_dof
    c5 = workload == "c5"
    u0 = np.zeros(n)
    u0[v2d] = gaussian_ic(a1, a2, deltax) if c5 else slotted_disc_ic(a1, a2, deltax)
    sb = otraj.SolidBody(asm, om=np.pi / 40, rot_scale=0.0 if c5 else 1.0)
    tl = (sample + 1) * n
    ck = np.ones(tl) if c5 else synthetic_control(mesh.x, mesh.y, v2d, sample)[:tl]
    uhat = (0.9 * np.tile(u0, sample + 1)) if c5 else np.roll(u0, 7)
    uk = np.zeros(tl)
    uk[:n] = u0
    t0 = time.perf_counter()
    otraj.solidbody_forward(sb, ck, uk, n, sample, dt)
    otraj.solidbody_adjoint(sb, ck, uk, uhat, np.zeros(tl), n, sample, dt, optim="alltime" if c5 else "finaltime")
    return {"seconds": time.perf_counter() - t0, "steps": 2 * sample, "beta_index": beta_index}


def cpu_baseline_sweep(n_proc, workload, sample):
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{workload},{sample},{k}"],
                              stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES=""))
             for k in range(n_proc)]
    t0 = time.perf_counter()
    # a worker that hangs must not hang the bench: the sample is sized for tens of seconds; past the limit it is killed and
    # reported (workers_ok < cores, 'error')
    limit = max(300.0, 4.0 * sample)
    outs, errors = [], []
    for k, p in enumerate(procs):
        try:
            outs.append(p.communicate(timeout=max(1.0, limit - (time.perf_counter() - t0)))[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate()[0])
            errors.append(f"worker {k}: killed after {limit:.0f} s")
            continue
        if p.returncode != 0:
            errors.append(f"worker {k}: exit code {p.returncode}")
    wall = time.perf_counter() - t0
    recs = [json.loads(o.strip().splitlines()[-1]) for o, p in zip(outs, procs) if p.returncode == 0 and o.strip()]
    steps = sum(r["steps"] for r in recs)
    slowest = max((r["seconds"] for r in recs), default=float("nan"))
    return {"value": steps / slowest if recs else None, "unit": "timesteps/s", "cores": n_proc, "kind": "port",
            "host_cpus": os.cpu_count(), "workers_ok": len(recs), **({"error": "; ".join(errors)} if errors else {}),
            "sample": f"{n_proc} concurrent 1-thread oracle processes, each {sample} forward + {sample} adjoint FCT steps of the "
                      f"{workload.upper()} workload (per-step assembly + vectorised NumPy/SciPy FCT step with SuperLU); "
                      "value = all steps / the slowest worker's time",
            "wall_s_incl_startup": wall}


def _profiled_forward(prob, ctx, d_c, d_u, steps):
    for _ in range(6):                         # warm-up: the sweep budget / launch plan settles (one trial of
        prob.forward(d_c, d_u, batch=1)        # fewer launches may fail and repeat a sweep on the way)
    ctx.synchronize()
    t0 = time.perf_counter()
    prob.forward(d_c, d_u, batch=1)            # graph replay, un-profiled: whole-step time
    ctx.synchronize()
    step_ms = 1e3 * (time.perf_counter() - t0) / steps
    ctx.set_profiling(True)
    prob.forward(d_c, d_u, batch=1)
    rep = ctx.profile_report()
    ctx.set_profiling(False)
    log = prob.solver_log(1)
    return step_ms, rep, int(log["solver_iters"].sum())


def _kernel_table(rep, bpr, n, units, traffic, one_sweep_units):
    """Per kernel class: compulsory bytes of a launch as executed / measured launch time."""
    kernels = {}
    for k, (ms, cnt) in rep.items():
        if not cnt or k not in bpr:
            continue
        per_launch = bpr[k] * n * (units.get(k, cnt) / cnt if k == "assemble" else 1.0)
        avg_ms = ms / cnt
        e = {"launches": cnt, "avg_launch_ms": avg_ms, "total_ms": ms,
             "compulsory_bytes_per_row_per_launch": per_launch / n,
             "achieved_GBps": per_launch / (1e6 * avg_ms), "frac": per_launch / (1e6 * avg_ms) / HBM_PEAK_GBS}
        if k in one_sweep_units and k in ONE_SWEEP_BYTES:
            e["sweeps_per_launch"] = one_sweep_units[k] / cnt
            e["one_sweep_equiv_GBps"] = ONE_SWEEP_BYTES[k] * n * one_sweep_units[k] / (1e6 * ms)
        if traffic and k in traffic:
            e["traffic_bytes_per_launch"] = traffic[k]
            e["traffic_bytes_per_row"] = traffic[k] / n
            # FETCH_SIZE counts Infinity-Cache hits as well (MI355X_MICROARCH.md, HBM): an upper bound on HBM bytes
            e["traffic_frac_incl_infinity_cache"] = traffic[k] / (1e6 * avg_ms) / HBM_PEAK_GBS
            e["overfetch"] = traffic[k] / per_launch
        kernels[k] = e
    return kernels


def roofline(hp, solvers, n_cells, steps, device_id):
    """HBM roofline on a mesh far larger than L2 + Infinity Cache: per-class HIP-event timing of one forward
    sweep (events recorded on the library's own stream around every launch)."""
    a1, a2 = -1.0, 1.0
    h = (a2 - a1) / n_cells
    dt = 1e-3 * h / 0.025                      # same CFL number as C2
    mesh = hp.SquareMeshP1(a1, a2, n_cells)
    n = mesh.nodes
    prob = solvers.SolidBodyDrift(mesh, steps, dt, batch=1, device_id=device_id, order=hp.ORDER_VERTEX)
    ctx = prob.ctx
    x, y = mesh.coordinates()
    rng = np.random.default_rng(0)
    u0 = np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n)
    c = np.tile(1.0 + 0.5 * np.sin(3 * x) * np.cos(2 * y), steps + 1)
    d_c = ctx.array(c)
    init = np.zeros((steps + 1) * n)
    init[:n] = u0
    d_u = ctx.array(init)
    regime = int(ctx.kernel_regime(1))
    geom = os.environ.get("FEMFCT_GEOM_MASS", "1") != "0"
    sha = source_sha16()

    # PMC-derived HBM bytes per launch: only when measured on this very source tree (tools/refresh_profiles.sh)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic, traffic1, traffic_note = None, None, "profiles/traffic.json absent"
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        ent = tj.get(f"n{n}")
        if ent is None:
            traffic_note = f"no PMC record for n={n}"
        elif ent.get("source_sha16") != sha:
            traffic_note = (f"PMC record is from kernel sources {ent.get('source_sha16')}, this build is {sha}: "
                            "omitted (re-run tools/refresh_profiles.sh)")
        else:
            traffic, traffic_note = ent["bytes_per_launch"], f"rocprofv3 PMC passes, source_sha16 {sha}"
            traffic1 = ent.get("one_sweep_bytes_per_launch")

    # ---- (1) the product path at this size (fused multi-sweep kernels)
    step_ms, rep, sweeps = _profiled_forward(prob, ctx, d_c, d_u, steps)
    fused_flux = rep["limit"][1] == 0
    inline_ops = rep["assemble"][1] == 0           # no stored operator: k_build_low_sb / k_dudt_rhs_sb
    half_d = regime == 3 and os.environ.get("FEMFCT_HALF_D", "1") != "0"
    l_nonzero = ctx.lowop_nonzero_fraction()       # 1.0 unless the zero mask is in force
    rot_geom = ctx.rotation_derived()              # Arot evaluated from the node positions in those two kernels, not read
    bpr = launch_bytes_per_row(fused=regime >= 2 and fused_flux, geom_mass=geom, inline_ops=inline_ops, half_d=half_d,
                               l_nonzero=l_nonzero, rot_geom=rot_geom)
    units = {"assemble": steps}
    one_sweep_units = {"jacobi": sweeps, "cheb": 19 * steps, "flux": steps}
    kernels = _kernel_table(rep, bpr, n, units, traffic, one_sweep_units)
    step_bytes = sum(e["compulsory_bytes_per_row_per_launch"] * n * e["launches"] for e in kernels.values()) / steps
    step_traffic = (sum(e["traffic_bytes_per_launch"] * e["launches"] for e in kernels.values()
                        if "traffic_bytes_per_launch" in e) / steps) if traffic else None
    dom_name = max(("jacobi", "cheb"), key=lambda k: kernels[k]["total_ms"])
    dom = kernels[dom_name]
    walkers = ctx.patch_walkers(1, sweeps // max(1, steps)) if regime == 3 else 0
    linfo = ctx.launch_info() if regime == 3 else {}
    sym = {3: {"jacobi": {"k_strip_jacobi_pair_walk": "k_strip_jacobi_pair_walk<6, 8>"}.get(linfo.get("jacobi_kernel"), linfo.get("jacobi_kernel", "k_strip4_jacobi<0>")),
               "cheb": (("k_strip4_cheb_mass_int + k_strip4_cheb_mass on the boundary ring" if os.environ.get("FEMFCT_T4_INT", "1") != "0"
                         else "k_strip4_cheb_mass_walk") if walkers else "k_strip4_cheb_mass") if geom else "k_strip4_cheb"},
           2: {"jacobi": "k_tile_jacobi<H,0,BIG>", "cheb": "k_tile_cheb<H>"},
           1: {"jacobi": "k_strip_jacobi<RPT>", "cheb": "k_strip_cheb<RPT>"},
           0: {"jacobi": "k_jacobi<7,256,1>", "cheb": "k_cheb<7,256,1>"}}[regime][dom_name]
    out = {"bound": "hbm", "kernel": f"{sym} ({dom.get('sweeps_per_launch', 1):.1f} sweeps per launch)",
           "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
           "traffic": dom.get("traffic_bytes_per_launch"), "traffic_source": traffic_note,
           "traffic_frac_incl_infinity_cache": dom.get("traffic_frac_incl_infinity_cache"),
           "one_sweep_equiv_GBps": dom.get("one_sweep_equiv_GBps"),
           # the same launch priced as if the vanishing entries of L had to be streamed (dense 7-slot rows: 80 B/row)
           "frac_counting_zero_entries": (ONE_SWEEP_BYTES[dom_name] * n / (1e6 * dom["avg_launch_ms"]) / HBM_PEAK_GBS
                                          if dom_name == "jacobi" else None),
           "definition": "achieved = compulsory bytes of one launch as executed (each array once; L once per multi-sweep "
                         "launch, without its exactly-zero off-diagonals, which are neither stored nor loaded) / mean "
                         "launch time by HIP events; frac = achieved / 8 TB/s",
           "workload": f"synthetic square mesh {n_cells + 1}x{n_cells + 1} (n={n}), vertex order, same CFL as C2",
           "compulsory_bytes_per_launch": dom["compulsory_bytes_per_row_per_launch"] * n,
           "avg_launch_ms": dom["avg_launch_ms"], "kernel_regime": regime, "patch_walkers": linfo.get("jacobi_walkers", walkers),
           "launch_info": linfo, "source_sha16": sha,
           "operator": ("derived inside k_build_low_sb / k_dudt_rhs_sb" + (" (rotation part from the node positions, no Arot read)" if rot_geom else ""))
                       if inline_ops else "stored by k_ops_solidbody",
           "low_order_offdiag_nonzero_fraction": l_nonzero, "d_stored_once_per_edge": half_d,
           "fct_step_ms": step_ms, "fct_steps_per_s": 1e3 / step_ms, "jacobi_sweeps_per_step": sweeps / steps,
           "step": {"compulsory_bytes": step_bytes, "compulsory_GBps": step_bytes / (1e6 * step_ms),
                    "compulsory_frac": step_bytes / (1e6 * step_ms) / HBM_PEAK_GBS,
                    "traffic_bytes": step_traffic,
                    "traffic_frac_incl_infinity_cache": (step_traffic / (1e6 * step_ms) / HBM_PEAK_GBS) if step_traffic else None},
           "kernels": kernels}

    # ---- (2) fusions off: the one-sweep SpMV-type (Jacobi / Chebyshev) and limiter kernels of the north star
    ctx.set_fusion(False, False)
    step1_ms, rep1, sweeps1 = _profiled_forward(prob, ctx, d_c, d_u, steps)
    bpr1 = launch_bytes_per_row(fused=False, geom_mass=False)
    k1 = {}
    for k, (ms, cnt) in rep1.items():
        if not cnt or k not in bpr1:
            continue
        # launches enqueued after the device-side convergence test has passed return at once: price the sweeps
        # that ran, over the time of all launches (conservative)
        ran = sweeps1 if k == "jacobi" else (steps if k == "assemble" else cnt)
        by = bpr1[k] * n * ran
        k1[k] = {"launches": cnt, "executed": ran, "total_ms": ms, "compulsory_bytes_per_row_per_launch": bpr1[k],
                 "achieved_GBps": by / (1e6 * ms), "frac": by / (1e6 * ms) / HBM_PEAK_GBS}
        if traffic1 and k in traffic1 and k != "assemble":
            k1[k]["traffic_bytes_per_row"] = traffic1[k] / n
            k1[k]["traffic_frac_incl_infinity_cache"] = traffic1[k] * ran / (1e6 * ms) / HBM_PEAK_GBS
    out["one_sweep_kernels"] = {"fct_step_ms": step1_ms, "fct_steps_per_s": 1e3 / step1_ms,
                                "jacobi_sweeps_per_step": sweeps1 / steps, "kernels": k1,
                                "note": "femfct_set_fusion(0, 0): k_jacobi / k_cheb (SpMV-type) and k_flux + k_limit, "
                                        "one matrix stream per launch"}
    prob.close()
    return out


def cpu_baseline(a1, a2, n_cells, Nt, dt, om, u0, ck, uhat, sample, gpu_u=None, gpu_p=None, rot_scale=1.0, optim="finaltime"):
    """CPU oracle (test infrastructure) timed on this host as the reported CPU baseline, and -- the oracle being
    the checker -- compared with the GPU trajectories the timed region produced for the same inputs."""
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import traj as otraj, fct as ofct
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(1)
    except Exception:  # pragma: no cover
        limiter = None
    mesh = SquareMesh(a1, a2, n_cells)
    asm = P1Assembler(mesh)
    n = mesh.nodes
    sb = otraj.SolidBody(asm, om=om, rot_scale=rot_scale)
    ns = min(sample, Nt)
    uk = np.zeros((ns + 1) * n)
    uk[:n] = u0
    c = ck[:(ns + 1) * n]
    t0 = time.perf_counter()
    otraj.solidbody_forward(sb, c, uk, n, ns, dt)
    pk = np.zeros_like(uk)
    otraj.solidbody_adjoint(sb, c, uk, uhat[:(ns + 1) * n] if optim == "alltime" else uhat, pk, n, ns, dt, optim=optim)
    t_vec = time.perf_counter() - t0
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    parity = None
    if gpu_u is not None:
        m = (ns + 1) * n
        parity = {"u_rel_l2": rel(gpu_u[:m], uk), "u_final_rel_l2": rel(gpu_u[ns * n:m], uk[ns * n:]),
                  "p_rel_l2": rel(gpu_p[:m], pk) if ns == Nt else None, "tol": 1e-6,
                  "levels_compared": ns + 1,
                  "what": "GPU state/adjoint trajectories of the timed run vs the CPU oracle on the same control, "
                          "initial condition and target (81x81, dt 1e-3)"}
        parity["ok"] = bool(parity["u_rel_l2"] < 1e-6 and (parity["p_rel_l2"] is None or parity["p_rel_l2"] < 1e-6))
    # reference-cost-profile variant (LIL + interpreter loops, the reference's data structures)
    nb = mesh.dof_neighbors()
    A = -(sb.A_u(c[n:2 * n]))
    t1 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        ofct.fct_step_lil(A, np.zeros(n), u0, dt, n, sb.cm.M, sb.cm.ML, nb)
    t_lil = (time.perf_counter() - t1) / reps
    if limiter is not None and hasattr(limiter, "unregister"):
        limiter.unregister()
    base = {"value": 2 * ns / t_vec, "unit": "timesteps/s", "cores": 1, "kind": "port",
            "sample": f"{ns} forward + {ns} adjoint FCT steps of the {'C5' if optim == 'alltime' else 'C2'} workload (per-step assembly + "
                      f"vectorised NumPy/SciPy FCT step with SuperLU), 1 thread",
            "host_cpus": os.cpu_count(),
            "reference_profile_variant": {"value": 1.0 / t_lil, "unit": "timesteps/s",
                                          "sample": f"{reps} FCT steps with LIL matrices + Python loops "
                                                    "(the reference's data structures), no assembly"}}
    return base, parity, (uk, pk, ns)


def tolerance_table(hp, solvers, device_id, n_cells, steps, c2_eval, tols=(1e-13, 1e-11, 1e-9)):
    """What the low-order solve's tolerance buys (the default, 1e-13 ||b||, is not changed): for each rel_tol the sweeps
    per step and FCT steps/s of the roofline mesh, and the parity of C2's 250 + 250-step trajectories against the oracle
    (c2_eval(rel_tol) -> dict).  The reference solves this system directly (spsolve, helpers.py:1782); the north-star
    bar on the solution is 1e-6."""
    rows = []
    mesh = hp.SquareMeshP1(-1.0, 1.0, n_cells)
    n = mesh.nodes
    dt = 1e-3 * (2.0 / n_cells) / 0.025
    prob = solvers.SolidBodyDrift(mesh, steps, dt, batch=1, device_id=device_id, order=hp.ORDER_VERTEX)
    ctx = prob.ctx
    try:
        x, y = mesh.coordinates()
        rng = np.random.default_rng(0)
        init = np.zeros((steps + 1) * n)
        init[:n] = np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n)
        d_c, d_u = ctx.array(np.tile(1.0 + 0.5 * np.sin(3 * x) * np.cos(2 * y), steps + 1)), ctx.array(init)
        for tol in tols:
            ctx.set_solver(rel_tol=tol)
            for _ in range(4):                       # the sweep budget settles to the tolerance (a changed budget is a
                prob.forward(d_c, d_u, batch=1)      # new graph: ~0.1 s of capture, not part of a step)
            ctx.synchronize()
            el = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                prob.forward(d_c, d_u, batch=1)
                ctx.synchronize()
                el = min(el, time.perf_counter() - t0)
            log = prob.solver_log(1)
            rows.append({"rel_tol": tol, "large_mesh_steps_per_s": steps / el,
                         "large_mesh_sweeps_per_step": float(log["solver_iters"].mean()),
                         "large_mesh_resid_max": float(log["solver_resid"].max()), "c2": c2_eval(tol)})
    finally:
        prob.close()
    return {"workload": f"synthetic {n_cells + 1}x{n_cells + 1} mesh (the roofline mesh) and C2 (81x81, 250 + 250 steps)",
            "default_rel_tol": 1e-13, "rows": rows,
            "note": "sweeps are issued in multi-sweep launches (8-12 per launch), so the counts move in those quanta; "
                    "large-mesh rate = best of 3 forward sweeps after the sweep budget has settled"}


if __name__ == "__main__":
    main()
