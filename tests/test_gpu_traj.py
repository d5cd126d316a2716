"""GPU parity tests of the structured-mesh assembly and the trajectory sweeps (-m gpu)."""
import importlib

import os

import numpy as np
import pytest
from scipy.sparse import csr_matrix

from helpers_golden import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    mod = importlib.import_module("fem-fct-pdeco_amd")
    mod.fct_helpers.VERBOSE = False
    return mod


@pytest.fixture(scope="module")
def solvers():
    return importlib.import_module("fem-fct-pdeco_amd.solvers")


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def ell_to_scipy(ctx, ell, n):
    """device ELL -> scipy CSR through the structured CSR map."""
    cols = ctx.ell_cols()
    W = ctx.W
    rows = np.tile(np.arange(n), W)
    mask = (cols.reshape(-1) != rows) | (np.arange(W * n) < n)
    nnz = int(mask.sum())
    vals = ctx.ell_to_csr(ell, nnz)
    # rebuild CSR pattern exactly as the library does: sorted columns per row
    r = rows[mask]
    c = cols.reshape(-1)[mask]
    order = np.lexsort((c, r))
    indptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))])
    return csr_matrix((vals, c[order], indptr), shape=(n, n))


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("nc,a1,a2", [(1, 0.0, 1.0), (4, -1.0, 1.0), (10, 0.0, 1.0), (40, -1.0, 1.0)])
def test_mesh_constants_and_convection(hp, order, nc, a1, a2):
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler, row_lump_diag
    from oracle.traj import rotation_wind, schnak_wind
    mesh = SquareMesh(a1, a2, nc)
    asm = P1Assembler(mesh)
    n = mesh.nodes
    ctx = hp.Context(0)
    ctx.set_mesh_square(a1, a2, nc, order)
    assert ctx.n == n and ctx.W == 7
    perm = mesh.vertex_to_dof if order == 0 else np.arange(n)   # oracle matrices are in DoF order

    def to_dof(Ad):  # device matrix (ctx ordering) -> DoF ordering
        if order == 1:
            return Ad
        P = csr_matrix((np.ones(n), (mesh.vertex_to_dof, np.arange(n))), shape=(n, n))
        return (P @ Ad @ P.T).tocsr()

    M = to_dof(ell_to_scipy(ctx, ctx.mass_ell, n))
    Ad = to_dof(ell_to_scipy(ctx, ctx.stiffness_ell, n))
    assert abs(M - asm.mass()).max() < 1e-14 * mesh.h ** 2
    assert abs(Ad - asm.stiffness()).max() < 1e-14
    ml = np.empty(n)
    hp._lib.check(ctx.handle, hp._lib.lib.femfct_memcpy_d2h(ctx.handle, ml.ctypes.data, ctx.lumped_mass, n * 8))
    ml_dof = ml if order == 1 else ml[mesh.dof_to_vertex]
    assert np.max(np.abs(ml_dof - row_lump_diag(asm.mass()))) < 1e-14 * mesh.h ** 2
    xq, yq = ctx.quad_points(nc)
    for wind in (rotation_wind(np.pi / 40), schnak_wind):
        wx, wy = wind(xq, yq)
        A = ctx.assemble_convection(np.stack([wx, wy], axis=1).reshape(-1))
        Ao = asm.convection(wind)
        assert abs(to_dof(ell_to_scipy(ctx, A, n)) - Ao).max() < 1e-13 * max(1.0, abs(Ao).max())
    # the rigid rotation in closed form (femfct_assemble_rotation): the same integral as the quadrature, to rounding
    Ar = ctx.assemble_rotation(40.0 / np.pi)
    Ao = asm.convection(rotation_wind(np.pi / 40))
    assert abs(to_dof(ell_to_scipy(ctx, Ar, n)) - Ao).max() < 1e-14 * max(1.0, abs(Ao).max())
    ctx.close()


def test_solidbody_sweeps_vs_reference_fct(hp, solvers):
    """20 forward + 20 adjoint steps produced by the REAL reference FCT_alg_ref
    (tests/golden/solidbody_traj_N21.npz)."""
    z = load("solidbody_traj_N21.npz")
    a1, a2, nc = z["geom"]
    Nt, dt = int(z["Nt"]), float(z["dt"])
    mesh = hp.SquareMeshP1(a1, a2, int(nc))
    n = mesh.nodes
    prob = solvers.SolidBodyDrift(mesh, Nt, dt, om=float(z["om"]))
    uk = np.zeros((Nt + 1) * n)
    uk[:n] = z["uk"][:n]
    out = prob.solve_state(z["ck"], uk)
    assert out is uk
    assert rel(uk, z["uk"]) < 1e-9
    log = prob.solver_log(1)
    assert not np.any(log["flags"] & hp.FLAG_SOLVER_BUDGET)
    pk = prob.solve_adjoint(z["ck"], uk, z["uhat"], np.zeros_like(uk))
    assert rel(pk, z["pk"]) < 1e-9
    # a second sweep replays cached graphs with the adapted budget: identical bits
    uk2 = np.zeros_like(uk)
    uk2[:n] = uk[:n]
    prob.solve_state(z["ck"], uk2)
    assert np.array_equal(uk2, uk)
    prob.close()


@pytest.mark.parametrize("order", [0, 1])
def test_solidbody_alltime_adjoint_and_gradient_vs_oracle(hp, solvers, order):
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import traj as otraj
    rng = np.random.default_rng(3)
    nc, Nt, dt, om, beta = 12, 10, 2e-3, np.pi / 40, 0.1
    omesh = SquareMesh(-1, 1, nc)
    asm = P1Assembler(omesh)
    n = omesh.nodes
    sb = otraj.SolidBody(asm, om=om)
    ck = 3 * rng.random((Nt + 1) * n)
    u0 = np.exp(-20 * ((omesh.x + 0.3) ** 2 + (omesh.y + 0.2) ** 2))[omesh.dof_to_vertex]
    uk_o = np.zeros((Nt + 1) * n)
    uk_o[:n] = u0
    otraj.solidbody_forward(sb, ck, uk_o, n, Nt, dt)
    uhat = uk_o * 0.8 + 0.01 * rng.random(uk_o.size)
    pk_o = otraj.solidbody_adjoint(sb, ck, uk_o, uhat, np.zeros_like(uk_o), n, Nt, dt, optim="alltime")
    dk_o = otraj.solidbody_descent_direction(sb, ck, uk_o, pk_o, beta, n, Nt)

    # the device works in its own ordering; permute at the boundary when order == VERTEX
    v2d = omesh.vertex_to_dof

    def to_dev(x):
        return x if order == 1 else x.reshape(-1, n)[:, v2d].reshape(-1)

    def from_dev(x):
        if order == 1:
            return x
        out = np.empty_like(x.reshape(-1, n))
        out[:, v2d] = x.reshape(-1, n)
        return out.reshape(-1)

    prob = solvers.SolidBodyDrift(hp.SquareMeshP1(-1, 1, nc), Nt, dt, om=om, order=order)
    uk = np.zeros((Nt + 1) * n)
    uk[:n] = to_dev(u0)
    prob.solve_state(to_dev(ck), uk)
    assert rel(from_dev(uk), uk_o) < 1e-9
    pk = prob.solve_adjoint(to_dev(ck), uk, to_dev(uhat), np.zeros_like(uk), optim="alltime")
    assert rel(from_dev(pk), pk_o) < 1e-9
    dk = prob.solve_descent_direction(to_dev(ck), uk, pk, beta)
    assert rel(from_dev(dk), dk_o) < 1e-9
    with pytest.raises(ValueError):
        prob.solve_adjoint(to_dev(ck), uk, to_dev(uhat), np.zeros_like(uk), optim="never")
    prob.close()


def test_batched_trajectories_match_single(hp, solvers):
    rng = np.random.default_rng(5)
    nc, Nt, dt, B = 10, 6, 2e-3, 4
    mesh = hp.SquareMeshP1(-1, 1, nc)
    n = mesh.nodes
    prob = solvers.SolidBodyDrift(mesh, Nt, dt, batch=B)
    tl = (Nt + 1) * n
    cks = 3 * rng.random((B, tl))
    u0 = rng.random(n)
    singles = []
    for b in range(B):
        uk = np.zeros(tl)
        uk[:n] = u0
        singles.append(prob.solve_state(cks[b], uk).copy())
    c = prob.ctx.array(cks.reshape(-1))
    u = prob.ctx.zeros(B * tl)
    init = np.zeros((B, tl))
    init[:, :n] = u0
    u.upload(init.reshape(-1))
    prob.forward(c, u, batch=B)
    out = u.download().reshape(B, tl)
    for b in range(B):
        assert np.array_equal(out[b], singles[b])
    J = prob.cost(u, c, c, 0.3, "alltime", batch=B)
    for b in range(B):
        Jb = hp.cost_functional(out[b], cks[b], cks[b], Nt, dt, _mass(hp, mesh), 0.3, "alltime")
        assert abs(J[b] - Jb) <= 1e-12 * abs(Jb)
    prob.close()


def _mass(hp, mesh):
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    return P1Assembler(SquareMesh(mesh.a1, mesh.a2, mesh.n_cells)).mass()


@pytest.mark.parametrize("optim", ["finaltime", "alltime"])
def test_pgd_solidbody_matches_oracle_loop_and_speculative_equals_sequential(hp, solvers, optim):
    """Projected gradient loop of advection_solidbody_FCT_PDECO_finaltime_Garvie.py:164-330 (and of
    ..._alltime_Garvie.py, config C5's loop) on the device vs the same loop written with the CPU oracle."""
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import traj as otraj, fct as ofct
    nc, Nt, dt, om, beta, lo, hi = 12, 6, 2e-3, np.pi / 40, 0.05, 0.0, 5.0
    gam, s0, max_armijo, iters = 1e-4, 1.0, 4, 2
    omesh = SquareMesh(-1, 1, nc)
    asm = P1Assembler(omesh)
    n = omesh.nodes
    tl = (Nt + 1) * n
    rng = np.random.default_rng(7)
    u0 = np.exp(-15 * ((omesh.x + 0.2) ** 2 + (omesh.y - 0.1) ** 2))[omesh.dof_to_vertex]
    uhat = np.exp(-15 * ((omesh.x + 0.1) ** 2 + (omesh.y - 0.2) ** 2))[omesh.dof_to_vertex]
    c0 = np.ones(tl)

    # ---- oracle loop
    sb = otraj.SolidBody(asm, om=om)
    if optim == "alltime":      # target trajectory: the oracle's own forward solve at the true control c = 2
        uhat = np.zeros(tl); uhat[:n] = u0
        otraj.solidbody_forward(sb, 2.0 * np.ones(tl), uhat, n, Nt, dt)
    uk, _, c_prev, h_o = otraj.solidbody_pgd_loop(sb, u0, uhat, c0, beta, lo, hi, iters, n, Nt, dt, gam, s0, max_armijo, optim)
    costs, ks = h_o["cost"], h_o["armijo_k"]

    prob = solvers.SolidBodyDrift(hp.SquareMeshP1(-1, 1, nc), Nt, dt, om=om)
    pgd = solvers.pgd_solidbody_finaltime if optim == "finaltime" else solvers.pgd_solidbody_alltime
    u_s, p_s, c_s, h_s = pgd(prob, u0, uhat, c0, beta, lo, hi, iters, gam, s0, max_armijo, True)
    u_q, p_q, c_q, h_q = pgd(prob, u0, uhat, c0, beta, lo, hi, iters, gam, s0, max_armijo, False)
    assert h_s["armijo_k"] == ks and h_q["armijo_k"] == ks
    assert np.allclose(h_s["cost"], costs, rtol=1e-9, atol=0)
    # the accept / reject decisions are far from their thresholds (on both sides the same margins)
    for ms_d, ms_o in zip(h_s["armijo_margin"], h_o["armijo_margin"]):
        assert np.allclose(ms_d, ms_o, rtol=1e-6, atol=1e-9)
    assert h_o["armijo_margin_min"] > 1e-7 and h_s["armijo_margin_min"] > 1e-7
    assert rel(c_s, c_prev) < 1e-8 and rel(u_s, uk) < 1e-8
    assert rel(c_s, c_q) < 1e-11 and rel(u_s, u_q) < 1e-11 and np.allclose(h_s["cost"], h_q["cost"], rtol=1e-11)
    prob.close()


@pytest.mark.parametrize("nc,order", [(40, 0), (40, 1), (330, 0), (767, 0), (1000, 0), (2303, 0)])
def test_fused_kernels_agree_with_one_sweep_kernels(hp, solvers, nc, order):
    """Strip / tile multi-sweep kernels (331^2, 768^2, 1001^2 nodes: 64-patch strip kernels with partial edge
    tiles and halo depths 8-10; 2304^2 nodes: the separate residual-reduce kernel) against the plain one-sweep
    kernels on the same inputs."""
    Nt = 3 if nc < 2000 else 1
    mesh = hp.SquareMeshP1(-1, 1, nc)
    n = mesh.nodes
    dt = 1e-3 * (2.0 / nc) / 0.05
    prob = solvers.SolidBodyDrift(mesh, Nt, dt, order=order)
    x, y = mesh.coordinates()
    rng = np.random.default_rng(11)
    u0 = np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n)
    c = np.tile(1.0 + 0.5 * np.sin(3 * x) * np.cos(2 * y), Nt + 1)
    outs = []
    for strips, tiles in ((False, False), (True, False), (True, True)):
        prob.ctx.set_fusion(strips, tiles)
        uk = np.zeros((Nt + 1) * n)
        uk[:n] = u0
        prob.solve_state(c, uk)
        log = prob.solver_log(1)
        assert not np.any(log["flags"] & hp.FLAG_SOLVER_BUDGET)
        outs.append(uk.copy())
    assert rel(outs[1], outs[0]) < 1e-11
    assert rel(outs[2], outs[0]) < 1e-11
    prob.close()


def test_large_batch_uses_bandwidth_tiles_and_matches_single(hp, solvers):
    """n * batch >= 90k switches to the 64 x 64-patch kernels (four nodes per thread)."""
    rng = np.random.default_rng(9)
    nc, Nt, dt, B = 80, 2, 1e-3, 64
    mesh = hp.SquareMeshP1(-1, 1, nc)
    n = mesh.nodes
    prob = solvers.SolidBodyDrift(mesh, Nt, dt, batch=B, order=hp.ORDER_VERTEX)
    tl = (Nt + 1) * n
    x, y = mesh.coordinates()
    u0 = np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n)
    cks = 1.0 + rng.random((B, tl))
    c = prob.ctx.array(cks.reshape(-1))
    init = np.zeros((B, tl))
    init[:, :n] = u0
    u = prob.ctx.array(init.reshape(-1))
    prob.forward(c, u, batch=B)
    out = u.download().reshape(B, tl)
    assert not np.any(prob.solver_log(B)["flags"] & hp.FLAG_SOLVER_BUDGET)
    for b in (0, 17, 63):
        uk = np.zeros(tl)
        uk[:n] = u0
        prob.solve_state(cks[b], uk)          # batch 1: latency-regime kernels
        assert rel(out[b], uk) < 1e-11
    prob.close()


@pytest.mark.parametrize("dt,launches", [(4e-4, 2), (1e-3, 3)])
def test_deferred_residual_test_reproduces_the_in_launch_one(hp, solvers, monkeypatch, dt, launches):
    """FEMFCT_DEFER_CHECK: in the latency regime a low-order solve of two to four launches no longer tests the previous
    launch's residual inside the next one; workgroup 0 of the du/dt kernel reduces all partials at once (solve_ctl.h).
    Same trajectories to the bit, same per-step solver records (sweeps, relative residual, minimal row sum, flags),
    forward and adjoint, both DoF orders; a batch of trajectories too.  dt = 4e-4: two launches; 1e-3: three."""
    nc, Nt = 80, 12
    mesh = hp.SquareMeshP1(-1, 1, nc)
    n = mesh.nodes
    x, y = mesh.coordinates()
    rng = np.random.default_rng(23)
    u0 = np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n)
    c = np.tile(1.0 + 0.5 * np.sin(3 * x) * np.cos(2 * y), Nt + 1)
    for order, batch in ((hp.ORDER_VERTEX, 1), (hp.ORDER_FENICS, 1), (hp.ORDER_VERTEX, 3)):
        res = []
        for defer in ("0", "1"):
            monkeypatch.setenv("FEMFCT_DEFER_CHECK", defer)
            prob = solvers.SolidBodyDrift(mesh, Nt, dt, batch=batch, order=order)
            try:
                tile_regime = prob.ctx.kernel_regime(batch) == 2      # (a tuning knob may have switched the tile kernels off)
                tl = (Nt + 1) * n
                init = np.zeros((batch, tl))
                init[:, :n] = u0 * (1.0 + 0.1 * np.arange(batch))[:, None]
                d_c = prob.ctx.array(np.tile(c, batch))
                d_u = prob.ctx.array(init.reshape(-1))
                for _ in range(3):                       # the sweep budget settles at two launches
                    prob.forward(d_c, d_u, batch=batch)
                logf = {k: v.copy() for k, v in prob.solver_log(batch).items()}
                uk = d_u.download()
                d_p = prob.ctx.array(np.zeros(batch * tl))
                d_uhat = prob.ctx.array(np.tile(0.9 * uk[:tl].reshape(Nt + 1, n)[-1] + 0.01, batch))
                for _ in range(3):
                    prob.adjoint(d_c, d_u, d_uhat, d_p, "finaltime", batch=batch)
                loga = {k: v.copy() for k, v in prob.solver_log(batch).items()}
                res.append((uk, d_p.download(), logf, loga))
            finally:
                prob.close()
        (u0_, p0_, lf0, la0), (u1_, p1_, lf1, la1) = res
        if os.environ.get("FEMFCT_DEEP_HALO", "1") == "0":
            # shallow halos (a tuning knob): the same sweeps now take one launch more, the last of which the in-launch test
            # may skip while the deferred one runs it (solve_ctl.h: "the later ones ran nevertheless") -- then the two agree
            # to the solver tolerance, not to the bit, and the deferred run never did fewer sweeps
            assert rel(u1_, u0_) < 1e-11 and rel(p1_, p0_) < 1e-11
            assert np.all(lf1["solver_iters"] >= lf0["solver_iters"]) and np.all(la1["solver_iters"] >= la0["solver_iters"])
            continue
        assert np.array_equal(u0_, u1_) and np.array_equal(p0_, p1_)
        for a, b in ((lf0, lf1), (la0, la1)):
            # two to four launches, all needed: the case the deferral covers (dt = 1e-3: 3 x 12 or 4 x 10 sweeps)
            worst = int(a["solver_iters"].max())
            # whatever kernels a knob selects, the operator needs 24-26 (dt = 4e-4) / 33-36 (dt = 1e-3) Jacobi sweeps; fused
            # launches report whole launches of 8-13 sweeps on top of that
            assert (18, 30)[launches > 2] <= worst <= (30, 52)[launches > 2], worst
            if tile_regime and os.environ.get("FEMFCT_DEEP_HALO", "1") != "0":
                assert 13 * (launches - 1) < worst <= 13 * (launches + (launches > 2)), worst
            for k in a:
                assert np.array_equal(a[k], b[k]), k


def test_graph_relative_levels_match_eager_stepping(hp, solvers, monkeypatch):
    """Trajectory sweeps replay captured graphs of (here) 10 steps in which step r carries its level offset r * delta and only the
    last step moves the device counters.  With graphs off every step is its own group (offset 0, counters move each step):
    same trajectories to the bit and same per-step solver records, for a step count that is not a multiple of 10, forward
    (delta = +1) and adjoint (delta = -1, all-time right-hand side)."""
    monkeypatch.setenv("FEMFCT_STEPS_PER_GRAPH", "10")      # two full graphs and one of three steps
    nc, Nt, dt = 40, 23, 1e-3
    mesh = hp.SquareMeshP1(-1, 1, nc)
    n = mesh.nodes
    x, y = mesh.coordinates()
    rng = np.random.default_rng(31)
    u0 = np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n)
    c = np.tile(1.0 + 0.5 * np.sin(3 * x) * np.cos(2 * y), Nt + 1) + 0.05 * rng.random((Nt + 1) * n)
    res = []
    for graphs in (True, False):
        prob = solvers.SolidBodyDrift(mesh, Nt, dt, order=hp.ORDER_VERTEX)
        try:
            prob.ctx.set_graphs(graphs)
            tl = (Nt + 1) * n
            init = np.zeros(tl)
            init[:n] = u0
            d_c, d_u = prob.ctx.array(c), prob.ctx.array(init)
            for _ in range(2):
                prob.forward(d_c, d_u, batch=1)
            logf = {k: v.copy() for k, v in prob.solver_log(1).items()}
            uk = d_u.download()
            d_p = prob.ctx.array(np.zeros(tl))
            d_uhat = prob.ctx.array(0.9 * uk + 0.01)
            for _ in range(2):
                prob.adjoint(d_c, d_u, d_uhat, d_p, "alltime", batch=1)
            loga = {k: v.copy() for k, v in prob.solver_log(1).items()}
            res.append((uk, d_p.download(), logf, loga))
        finally:
            prob.close()
    (u_g, p_g, lf_g, la_g), (u_e, p_e, lf_e, la_e) = res
    assert np.array_equal(u_g, u_e) and np.array_equal(p_g, p_e)
    assert np.abs(u_g[-n:]).max() > 0 and np.abs(p_g[:n]).max() > 0
    for a, b in ((lf_g, lf_e), (la_g, la_e)):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


def test_walking_launches_with_a_batch(hp, solvers, monkeypatch):
    """The persistent-workgroup launches of the bandwidth regime split their walkers over the batch members (blockIdx.z);
    each member has its own partials, masks and carry.  Two trajectories in one batch at 331^2 with six walkers forced
    (three per member): identical bits to the same batch with one workgroup per patch, and equal to the trajectories
    solved one at a time up to the solver tolerance (a batch shares its sweep budget, a single solve has its own)."""
    nc, Nt = 330, 2
    mesh = hp.SquareMeshP1(-1.0, 1.0, nc)
    n = mesh.nodes
    dt = 1e-3 * (2.0 / nc) / 0.025
    x, y = mesh.coordinates()
    rng = np.random.default_rng(37)
    tl = (Nt + 1) * n
    u0 = [np.exp(-20 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.01 * rng.random(n),
          np.exp(-15 * ((x - 0.2) ** 2 + (y + 0.1) ** 2)) + 0.02 * rng.random(n)]
    cs = [np.tile(1.0 + 0.5 * np.sin(3 * x) * np.cos(2 * y), Nt + 1), np.tile(0.7 + 0.3 * np.cos(2 * x + y), Nt + 1)]
    init = np.zeros((2, tl))
    init[0, :n], init[1, :n] = u0
    outs = []
    for walk in ("1", "0"):
        monkeypatch.setenv("FEMFCT_T4_WALK", walk)
        monkeypatch.setenv("FEMFCT_T4_WALKERS", "6")
        prob = solvers.SolidBodyDrift(mesh, Nt, dt, batch=2, order=hp.ORDER_VERTEX)
        try:
            if walk == "1" and prob.ctx.uses_bandwidth_tiles(2) and os.environ.get("FEMFCT_T4_DPP", "1") != "0":   # (a tuning knob may have switched the tile / register-strip kernels off)
                assert prob.ctx.patch_walkers(2) == 3 and prob.ctx.patch_walkers(1) == 6
            else:
                assert prob.ctx.patch_walkers(2) == 0
            d_c, d_u = prob.ctx.array(np.concatenate(cs)), prob.ctx.array(init.reshape(-1))
            prob.forward(d_c, d_u, batch=2)
            assert not np.any(prob.solver_log(2)["flags"] & hp.FLAG_SOLVER_BUDGET)
            outs.append(d_u.download().reshape(2, tl))
            if walk == "1":
                for b in range(2):
                    uk = np.zeros(tl)
                    uk[:n] = u0[b]
                    prob.solve_state(cs[b], uk)
                    assert rel(uk, outs[0][b]) < 1e-11, b
        finally:
            prob.close()
    assert np.array_equal(outs[0], outs[1])
