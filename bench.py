#!/usr/bin/env python3
"""Benchmark of the FEM-FCT forward+adjoint hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N > 1 launched by
``python -m torch.distributed.run --nproc-per-node N ...``, one rank per GPU) prints ONE
JSON line on rank 0.

Workload (BASELINE.json configs[1], "C2" of SURVEY.md 8d): solid-body rotation + drift
control on [-1,1]^2, 81x81 P1 nodes (dx = 0.025, n = 6561), dt = 1e-3, T = 0.25 (250 steps),
eps = 0, rotation (-y,x)*40/pi, drift b = (1,1), slotted-disc initial condition.
One bench "step" = one cost + gradient evaluation of the projected-gradient loop:
forward sweep (250 FCT steps, per-step on-device assembly of the drift matrices),
cost functional, adjoint sweep (250 FCT steps), descent direction (251 Chebyshev solves).
Inputs (control, initial condition, target) are resident in HBM before the timed region.
metric = FCT timesteps (forward + adjoint) per second, whole job.

N > 1: every rank runs the same sweep for its own regularisation value beta (config C5:
embarrassingly parallel), and the ranks all-gather their cost values over RCCL once per
step; "scaling": "weak".

Extra objects in the JSON line:
  roofline      HBM roofline of the dominant kernel (Chebyshev/SpMV step), measured on a
                large synthetic mesh (n = 2049^2: the C2 working set is cache resident)
                with HIP events on the library's stream; all step kernels in "kernels".
  cpu_baseline  the CPU oracle (reference-faithful NumPy/SciPy restatement, SuperLU) timed
                on this host, 1 core, on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

# compulsory HBM bytes per matrix row and launch of each step kernel (ELL width 7, float64
# values, int32 column indices for the 6 off-diagonal slots)
BYTES_PER_ROW = {
    "build_low": 7 * 8 + 6 * (4 + 1) + 7 * 8 + 7 * 8 + 8 * 4,   # A, cols+tslot, L, D, ml/u/b/x0
    "jacobi": 7 * 8 + 6 * 4 + 8 * 3,                            # L, cols, b/x_in/x_out
    "dudt_rhs": 7 * 8 + 6 * 4 + 8 * 5,                          # A, cols, x, M_diag, u_L, r, y1
    "cheb": 7 * 8 + 6 * 4 + 8 * 4,                              # M, cols, b/y_mid/y_old/y_new
    "flux": 6 * 8 * 3 + 6 * 4 + 8 * 5,                          # M, D, F(write), cols, u/du/ml/R+/R-
    "limit": 6 * 8 + 6 * 4 + 8 * 5,                             # F, cols, R+/R-/u_L/ml/out
    "assemble": 7 * 8 * 3 + 6 * 4 + 8,                          # Ad, Arot, A(write), cols, c
}


def slotted_disc_ic(a1, a2, deltax, slit=0.05):
    """advection_solidbody_FCT_PDECO_finaltime.py:71-88 (np.arange grid, vertex order)."""
    X = np.arange(a1, a2 + deltax, deltax)
    X, Y = np.meshgrid(X, X)
    R = np.sqrt(X ** 2 + (Y - 1 / 3) ** 2)
    return ((R < 1 / 3) & ((np.abs(X) > slit) | (Y > 0.5))).astype(np.float64).reshape(-1)


def synthetic_control(mesh, num_steps, seed=0):
    """Smooth space-time control in the admissible box [0,5] (synthetic data)."""
    x, y = mesh.coordinates()
    v2d = mesh.vertex_to_dof
    t = np.linspace(0.0, 1.0, num_steps + 1)[:, None]
    c = 1.5 + 1.0 * np.sin(2 * np.pi * (x[None, :] + t)) * np.cos(np.pi * y[None, :]) + 0.5 * t
    out = np.empty_like(c)
    out[:, v2d] = c
    return np.clip(out, 0.0, 5.0).reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="independent trajectories per launch on each GPU")
    ap.add_argument("--roofline-cells", type=int, default=2048, help="cells per side of the roofline mesh (0: skip)")
    ap.add_argument("--roofline-steps", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=250,
                    help="forward+adjoint oracle steps each; 250 = the whole C2 sweep, ~8 s of one core (0: skip)")
    ap.add_argument("--pgd-iters", type=int, default=5, help="projected-gradient iterations of the C2 problem (0: skip)")
    ap.add_argument("--batched", type=str, default="8,64", help="extra batch sizes reported in 'batched' ('' : skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm

    hp = importlib.import_module("fem-fct-pdeco_amd")
    solvers = importlib.import_module("fem-fct-pdeco_amd.solvers")
    hp.fct_helpers.VERBOSE = False

    # ------------------------------------------------------------ workload C2
    a1, a2, deltax, dt, T = -1.0, 1.0, 0.1 / 2 / 2, 0.001, 0.25
    n_cells = round((a2 - a1) / deltax)
    Nt = round(T / dt)
    om = np.pi / 40
    B = args.batch
    mesh = hp.SquareMeshP1(a1, a2, n_cells)
    n = mesh.nodes
    tl = (Nt + 1) * n
    betas = [10.0 ** (-k / 2) for k in range(8)]          # C5 sweep values
    beta = betas[rank % len(betas)] if world > 1 else 1.0  # C2: beta = 1
    # device arrays live in dolfin vertex order (the library's fast layout: index-free stencil
    # addressing + 2-D tile kernels); the DoF-ordered arrays below feed the CPU oracle
    prob = solvers.SolidBodyDrift(mesh, Nt, dt, om=om, eps=0.0, batch=B, device_id=local_rank,
                                  order=hp.ORDER_VERTEX)
    ctx = prob.ctx
    to_dev = lambda x: hp.reorder_vector_from_dof(x, x.size // n, n, mesh.vertex_to_dof)
    u0 = hp.reorder_vector_to_dof(slotted_disc_ic(a1, a2, deltax), 1, n, mesh.vertex_to_dof)
    ck = synthetic_control(mesh, Nt, seed=rank)
    gpath = os.path.join(ROOT, "tests", "golden", "solidbody_t0.25_u.npz")
    uhat = np.load(gpath)["u"] if os.path.exists(gpath) else np.roll(u0, 7)
    init = np.zeros((B, tl))
    init[:, :n] = u0
    init[:, :n] = to_dev(u0)
    d_c = ctx.array(np.tile(to_dev(ck), B))
    d_u = ctx.array(init.reshape(-1))
    d_p = ctx.zeros(B * tl)
    d_d = ctx.zeros(tl)
    d_rhs = ctx.empty(tl)
    d_uhat = ctx.array(np.tile(to_dev(uhat), B))

    def one_step():
        prob.forward(d_c, d_u, batch=B)
        J = prob.cost(d_u, d_uhat, d_c, beta, "finaltime", batch=B)
        prob.adjoint(d_c, d_u, d_uhat, d_p, "finaltime", batch=B)
        prob.descent_direction(d_c, d_u, d_p, beta, d_d, scratch=d_rhs)   # batch member 0 (one control)
        if dist is not None:
            t_j = torch.tensor([float(J[0])], dtype=torch.float64, device=f"cuda:{local_rank}")
            out = [torch.empty_like(t_j) for _ in range(world)]
            dist.all_gather(out, t_j)                                     # RCCL: the sweep's only exchange
            return [float(o.item()) for o in out]
        return [float(J[0])]

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Js = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t_el = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        elapsed = float(t_el.item())
    log = prob.solver_log(B)
    fct_steps_per_bench_step = 2 * Nt * B * world
    value = fct_steps_per_bench_step * args.steps / elapsed

    result = {
        "metric": "FCT timesteps/sec (fwd+adj)", "value": value, "unit": "timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2 advection_solidbody_FCT_PDECO_finaltime: [-1,1]^2 81x81 P1 (dx=0.025, n=6561), "
                               "dt=1e-3, 250 fwd + 250 adj FCT steps per cost+gradient evaluation",
                   "nodes": n, "num_steps": Nt, "dt": dt, "batch_per_gpu": B,
                   "parallelism": f"beta-sweep x{world}" if world > 1 else "single trajectory",
                   "low_order_solver": "jacobi", "jacobi_sweeps_max": int(log["solver_iters"].max()),
                   "solver_resid_max": float(log["solver_resid"].max()), "graphs": True},
        "cost": Js[0],
    }

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
        result["roofline"] = roofline(hp, solvers, args.roofline_cells, args.roofline_steps, local_rank)
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
                prob.cost(ub, uhb, cb, beta, "finaltime", batch=Bx)
                prob.adjoint(cb, ub, uhb, pb, "finaltime", batch=Bx)

            for _ in range(2):
                sweep()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                sweep()
            ctx.synchronize()
            el = (time.perf_counter() - t0) / 2
            result["batched"].append({"batch_per_gpu": Bx, "value": 2 * Nt * Bx / el, "unit": "timesteps/s",
                                      "ms_per_step": 1e3 * el})
            for a in (cb, ub, pb, uhb):
                a.free()
    if rank == 0 and world == 1 and args.pgd_iters > 0:
        # the full optimisation loop of configs[1] (finaltime_Garvie.py:164-330), everything in HBM;
        # speculative = all 10 Armijo trial steps as one batch of independent trajectories
        pg = {}
        for spec in (True, False):
            t0 = time.perf_counter()
            _, _, _, hist = solvers.pgd_solidbody_finaltime(prob, to_dev(u0), to_dev(uhat), np.ones(tl), 1.0, 0.0, 5.0,
                                                            args.pgd_iters, speculative=spec)
            dt_it = (time.perf_counter() - t0) / len(hist["cost"])
            pg["speculative" if spec else "sequential"] = {
                "s_per_pgd_iteration": dt_it, "armijo_trials": hist["armijo_k"], "cost": hist["cost"][-1]}
        pg["cost_rel_diff"] = abs(pg["speculative"]["cost"] - pg["sequential"]["cost"]) / abs(pg["sequential"]["cost"])
        result["pgd_c2"] = pg
    if rank == 0 and world == 1 and args.cpu_sample > 0:      # the CPU baseline is an N = 1 figure
        result["cpu_baseline"] = cpu_baseline(a1, a2, n_cells, Nt, dt, om, u0, ck, uhat, args.cpu_sample)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def roofline(hp, solvers, n_cells, steps, device_id):
    """HBM roofline on a mesh far larger than L2 + Infinity Cache (per-class HIP-event timing of
    one forward sweep).  `achieved` follows the contract: algorithmic bytes of the work a launch
    performs (bytes/row of the one-sweep formulation x rows x sweeps in the launch) / launch time.
    The tile-fused Jacobi/Chebyshev kernels run 8 sweeps per pass over the matrix, so their real
    HBM traffic (`traffic`, from rocprofv3 PMC passes) is far below the algorithmic figure and
    `frac` can exceed 1; `hbm_frac` = traffic / time / peak is the physical utilisation."""
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
    sweeps = int(log["solver_iters"].sum())
    # sweeps / iterations / levels actually executed (the operators of all levels are assembled in one launch)
    units = {"jacobi": sweeps, "cheb": 19 * steps, "assemble": steps}
    fused_flux = rep["limit"][1] == 0                       # flux + limit in one launch
    bpr = dict(BYTES_PER_ROW)
    if fused_flux:
        bpr["flux"] = BYTES_PER_ROW["flux"] + BYTES_PER_ROW["limit"]
    tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch
    tr = json.load(open(tpath)).get(f"n{n}", {}) if os.path.exists(tpath) else {}
    kernels = {}
    for k, (ms, cnt) in rep.items():
        if not cnt or k not in bpr:
            continue
        u = units.get(k, cnt)
        alg = bpr[k] * n * u
        e = {"total_ms": ms, "launches": cnt, "avg_launch_ms": ms / cnt, "sweeps_per_launch": u / cnt,
             "bytes_per_row_per_sweep": bpr[k], "achieved_GBps": alg / (1e6 * ms), "frac": alg / (1e6 * ms) / HBM_PEAK_GBS}
        if k in tr:
            e["traffic_bytes_per_launch"] = tr[k]
            e["hbm_frac"] = tr[k] / (1e6 * ms / cnt) / HBM_PEAK_GBS
        kernels[k] = e
    dom_name = max(("jacobi", "cheb"), key=lambda k: kernels[k]["total_ms"])
    dom = kernels[dom_name]
    if dom["sweeps_per_launch"] > 1.5:
        sym = {"jacobi": "k_strip4_jacobi<0>", "cheb": "k_strip4_cheb"}[dom_name] if n >= 90000 else \
              {"jacobi": "k_tile_jacobi<H,0,0>", "cheb": "k_tile_cheb<H>"}[dom_name]
        label = f"{sym} (tile-fused, {dom['sweeps_per_launch']:.1f} sweeps per launch)"
    else:
        label = {"jacobi": "k_jacobi<7,256,*>", "cheb": "k_cheb<7,256,*>"}[dom_name] + " (one sweep per launch)"
    out = {"bound": "hbm", "kernel": label,
           "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
           "traffic": dom.get("traffic_bytes_per_launch"), "hbm_frac": dom.get("hbm_frac"),
           "workload": f"synthetic square mesh {n_cells + 1}x{n_cells + 1} (n={n}), vertex order, same CFL as C2",
           "algorithmic_bytes_per_launch": dom["bytes_per_row_per_sweep"] * n * dom["sweeps_per_launch"],
           "avg_launch_ms": dom["avg_launch_ms"], "fct_step_ms": step_ms, "fct_steps_per_s": 1e3 / step_ms,
           "jacobi_sweeps_per_step": sweeps / steps, "kernels": kernels}
    prob.close()
    return out


def cpu_baseline(a1, a2, n_cells, Nt, dt, om, u0, ck, uhat, sample):
    """CPU oracle (test infrastructure) timed on this host as the reported CPU baseline."""
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
    sb = otraj.SolidBody(asm, om=om)
    ns = min(sample, Nt)
    uk = np.zeros((ns + 1) * n)
    uk[:n] = u0
    c = ck[:(ns + 1) * n]
    t0 = time.perf_counter()
    otraj.solidbody_forward(sb, c, uk, n, ns, dt)
    pk = np.zeros_like(uk)
    otraj.solidbody_adjoint(sb, c, uk, uhat, pk, n, ns, dt, optim="finaltime")
    t_vec = time.perf_counter() - t0
    # reference-cost-profile variant (LIL + interpreter loops, the reference's data structures)
    nb = mesh.dof_neighbors()
    A = -(sb.A_u(c[n:2 * n]))
    t1 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        ofct.fct_step_lil(A, np.zeros(n), u0, dt, n, sb.cm.M, sb.cm.ML, nb)
    t_lil = (time.perf_counter() - t1) / reps
    if limiter is not None:
        limiter.unregister() if hasattr(limiter, "unregister") else None
    return {"value": 2 * ns / t_vec, "unit": "timesteps/s", "cores": 1, "kind": "port",
            "sample": f"{ns} forward + {ns} adjoint FCT steps of the C2 workload (per-step assembly + "
                      f"vectorised NumPy/SciPy FCT step with SuperLU), 1 thread",
            "host_cpus": os.cpu_count(),
            "reference_profile_variant": {"value": 1.0 / t_lil, "unit": "timesteps/s",
                                          "sample": f"{reps} FCT steps with LIL matrices + Python loops "
                                                    "(the reference's data structures), no assembly"}}


if __name__ == "__main__":
    main()
