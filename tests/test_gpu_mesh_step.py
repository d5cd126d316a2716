"""The one-workgroup-per-trajectory step (csrc/kernels_mesh.hip: meshes with N <= 42 nodes per side in vertex order, the
default there; 81 x 81 from 64 trajectories per launch on) through the C ABI (-m gpu): against the tile path it replaces (FEMFCT_MESH_STEP=0) on the same inputs,
against the CPU oracle, and its own invariants (batched == single bitwise, graphs on/off bitwise).  The systems of
configs 3 / 4 meet the oracle through this kernel in tests/test_gpu_fullsize.py and tests/test_gpu_systems.py."""
import importlib

import numpy as np
import pytest

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


def _inputs(hp, nc, Nt, B, seed):
    mesh = hp.SquareMeshP1(-1.0, 1.0, nc)
    n = mesh.nodes
    tl = (Nt + 1) * n
    rng = np.random.default_rng(seed)
    x, y = mesh.coordinates()
    u0 = np.exp(-10 * ((x + 0.3) ** 2 + (y - 0.2) ** 2)) + 0.05 * rng.random(n)
    c = 2.0 * rng.random((B, tl))                     # rough controls: rows of every kind
    src = 0.3 * rng.random((B, tl))
    return mesh, n, tl, u0, c, src


def _run(hp, solvers, monkeypatch, mesh_step, nc, Nt, B, eps, seed=7, graphs=True, large_from=None):
    monkeypatch.setenv("FEMFCT_MESH_STEP", "1" if mesh_step else "0")
    monkeypatch.setenv("FEMFCT_MESH_STEP_BATCH", "1")       # (the default; a knob-matrix run may have moved it)
    if large_from is None:
        monkeypatch.delenv("FEMFCT_MESH_STEP_BATCH_LARGE", raising=False)
    else:
        monkeypatch.setenv("FEMFCT_MESH_STEP_BATCH_LARGE", str(large_from))
    mesh, n, tl, u0, c, src = _inputs(hp, nc, Nt, B, seed)
    dt = 1e-3 * 80 / nc
    prob = solvers.SolidBodyDrift(mesh, Nt, dt, eps=eps, batch=B, order=hp.ORDER_VERTEX)
    ctx = prob.ctx
    try:
        ctx.set_graphs(graphs)
        regime = ctx.kernel_regime(B)
        init = np.zeros((B, tl))
        init[:, :n] = u0
        dc, du, dp, ds = ctx.array(c.ravel()), ctx.array(init.ravel()), ctx.zeros(B * tl), ctx.array(src.ravel())
        prob.forward(dc, du, batch=B, src=ds)
        log = prob.solver_log(B)
        u = du.download().reshape(B, tl)
        duh = ctx.array((0.9 * u + 0.01).ravel())
        prob.adjoint(dc, du, duh, dp, "alltime", batch=B)
        p = dp.download().reshape(B, tl)
    finally:
        prob.close()
    return regime, u, p, log


@pytest.mark.parametrize("eps", [0.0, 2e-3])
@pytest.mark.parametrize("nc", [4, 11, 25, 40, 41])
def test_mesh_step_matches_the_tile_path(hp, solvers, monkeypatch, nc, eps):
    """Forward sweep with a source term + all-time adjoint, 3 trajectories per launch, random controls, with and without
    physical diffusion; N = 5, 12 (even: every node of every 2 x 2 block in the mesh), 26, 41 (config meshes: own
    instantiation), 42 (the largest that fits).  The two paths solve the low-order system to the same tolerance with
    different iterations (block Gauss-Seidel vs Jacobi)."""
    Nt, B = 6, 3
    r1, u1, p1, log1 = _run(hp, solvers, monkeypatch, True, nc, Nt, B, eps)
    r0, u0, p0, log0 = _run(hp, solvers, monkeypatch, False, nc, Nt, B, eps)
    assert r1 == hp._lib.REGIME_MESH and r0 != hp._lib.REGIME_MESH
    assert np.isfinite(u1).all() and np.isfinite(p1).all()
    assert rel(u1, u0) < 1e-11 and rel(p1, p0) < 1e-10, (rel(u1, u0), rel(p1, p0))
    assert not np.any(log1["flags"] & hp.FLAG_SOLVER_BUDGET)
    assert log1["solver_resid"].max() <= 1e-13
    # the row-sum diagnostic of helpers.py:1796-1809 (sum_j L_ij from A's row sums here, from L's entries there)
    np.testing.assert_allclose(log1["min_rowsum"], log0["min_rowsum"], rtol=1e-9)
    assert np.array_equal(log1["flags"] & hp.FLAG_MMATRIX_ROWSUM, log0["flags"] & hp.FLAG_MMATRIX_ROWSUM)


@pytest.mark.parametrize("eps", [0.0, 2e-3])
def test_mesh_step_81x81_matches_the_tile_path(hp, solvers, monkeypatch, eps):
    """The 3 x 3-block instantiation for the 81 x 81 meshes of config C2 (the host picks it from 64 trajectories per
    launch on; forced on here for 3): same comparison as above."""
    nc, Nt, B = 80, 6, 3
    r1, u1, p1, log1 = _run(hp, solvers, monkeypatch, True, nc, Nt, B, eps, large_from=1)
    r0, u0, p0, log0 = _run(hp, solvers, monkeypatch, False, nc, Nt, B, eps)
    assert r1 == hp._lib.REGIME_MESH and r0 != hp._lib.REGIME_MESH
    assert np.isfinite(u1).all() and np.isfinite(p1).all()
    assert rel(u1, u0) < 1e-11 and rel(p1, p0) < 1e-10, (rel(u1, u0), rel(p1, p0))
    assert not np.any(log1["flags"] & hp.FLAG_SOLVER_BUDGET)
    assert log1["solver_resid"].max() <= 1e-13
    np.testing.assert_allclose(log1["min_rowsum"], log0["min_rowsum"], rtol=1e-9)


def test_mesh_step_81x81_is_the_default_from_64_trajectories_on(hp, solvers, monkeypatch):
    """Default selection at 81 x 81: tiles below 64 trajectories per launch, one workgroup per trajectory from 64 on
    (64 workgroups of 768 threads in one launch) -- against the tile path on the same 64 trajectories."""
    nc, Nt, B = 80, 4, 64
    monkeypatch.setenv("FEMFCT_MESH_STEP", "1")
    monkeypatch.delenv("FEMFCT_MESH_STEP_BATCH_LARGE", raising=False)
    mesh = hp.SquareMeshP1(-1.0, 1.0, nc)
    prob = solvers.SolidBodyDrift(mesh, Nt, 1e-3, batch=B, order=hp.ORDER_VERTEX)
    try:
        assert prob.ctx.kernel_regime(8) != hp._lib.REGIME_MESH
        assert prob.ctx.kernel_regime(63) != hp._lib.REGIME_MESH
        assert prob.ctx.kernel_regime(64) == hp._lib.REGIME_MESH
    finally:
        prob.close()
    _, ub, pb, logb = _run(hp, solvers, monkeypatch, True, nc, Nt, B, 0.0, seed=11)
    assert np.isfinite(ub).all() and not np.any(logb["flags"] & hp.FLAG_SOLVER_BUDGET)
    _, ut, pt, _ = _run(hp, solvers, monkeypatch, False, nc, Nt, B, 0.0, seed=11)
    assert rel(ub, ut) < 1e-11 and rel(pb, pt) < 1e-10, (rel(ub, ut), rel(pb, pt))


def test_mesh_step_batched_equals_single_and_graphs_are_neutral(hp, solvers, monkeypatch):
    nc, Nt, B = 40, 5, 4
    _, ub, pb, _ = _run(hp, solvers, monkeypatch, True, nc, Nt, B, 0.0, seed=3)
    _, ue, pe, _ = _run(hp, solvers, monkeypatch, True, nc, Nt, B, 0.0, seed=3, graphs=False)
    assert np.array_equal(ub, ue) and np.array_equal(pb, pe)             # captured graphs vs kernel-by-kernel
    monkeypatch.setenv("FEMFCT_MESH_STEP", "1")
    monkeypatch.setenv("FEMFCT_MESH_STEP_BATCH", "1")
    mesh, n, tl, u0, c, src = _inputs(hp, nc, Nt, B, 3)
    prob = solvers.SolidBodyDrift(mesh, Nt, 1e-3 * 80 / nc, batch=1, order=hp.ORDER_VERTEX)
    try:
        ctx = prob.ctx
        for b in range(B):                                               # one trajectory per launch: the same bits
            init = np.zeros(tl)
            init[:n] = u0
            dc, du, ds = ctx.array(c[b]), ctx.array(init), ctx.array(src[b])
            prob.forward(dc, du, batch=1, src=ds)
            assert np.array_equal(du.download(), ub[b])
    finally:
        prob.close()


def test_mesh_step_rowsum_flag_beyond_the_dt_restriction(hp, solvers, monkeypatch):
    """A time step far beyond the scheme's restriction: some row sum of L is not positive (the reference's "3: False"
    line, helpers.py:1796-1799) -- flagged by both paths on the same steps."""
    flags = []
    for mesh_step in (True, False):
        monkeypatch.setenv("FEMFCT_MESH_STEP", "1" if mesh_step else "0")
        mesh = hp.SquareMeshP1(-1.0, 1.0, 20)
        n = mesh.nodes
        prob = solvers.SolidBodyDrift(mesh, 1, 0.5, batch=1, order=hp.ORDER_VERTEX)
        try:
            ctx = prob.ctx
            x, y = mesh.coordinates()
            init = np.zeros(2 * n)
            init[:n] = np.exp(-10 * (x ** 2 + y ** 2))
            c = np.tile(5.0 * np.sin(3 * x) * np.cos(2 * y), 2)
            du = ctx.array(init)
            try:
                prob.forward(ctx.array(c), du, batch=1)
            except hp._lib.FemFctError:
                pass                                                     # (the solve itself may refuse such a step)
            log = prob.solver_log(1)
            flags.append(int(log["flags"][0, 0] & hp.FLAG_MMATRIX_ROWSUM))
            assert log["min_rowsum"][0, 0] <= 0.0
        finally:
            prob.close()
    assert flags == [hp.FLAG_MMATRIX_ROWSUM, hp.FLAG_MMATRIX_ROWSUM]


def test_mesh_step_vs_oracle_41x41(hp, solvers, monkeypatch):
    """Solid body on the 41 x 41 mesh, 25 forward + 25 all-time adjoint steps, against the CPU oracle (FEniCS DoF order
    there, vertex order here)."""
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import traj as otraj
    monkeypatch.setenv("FEMFCT_MESH_STEP", "1")
    monkeypatch.setenv("FEMFCT_MESH_STEP_BATCH", "1")
    nc, Nt, dt, om = 40, 25, 2e-3, np.pi / 40
    omesh = SquareMesh(-1.0, 1.0, nc)
    asm = P1Assembler(omesh)
    n = omesh.nodes
    rng = np.random.default_rng(5)
    sb = otraj.SolidBody(asm, om=om)
    v2d = omesh.vertex_to_dof
    ck = 3 * rng.random((Nt + 1) * n)                                    # DoF order
    u0 = (np.exp(-20 * ((omesh.x + 0.3) ** 2 + (omesh.y + 0.2) ** 2)))[omesh.dof_to_vertex]
    uo = np.zeros((Nt + 1) * n)
    uo[:n] = u0
    otraj.solidbody_forward(sb, ck, uo, n, Nt, dt)
    uhat = 0.8 * uo + 0.01 * rng.random(uo.size)
    po = otraj.solidbody_adjoint(sb, ck, uo, uhat, np.zeros_like(uo), n, Nt, dt, optim="alltime")
    to_dev = lambda a: np.ascontiguousarray(a.reshape(-1, n)[:, v2d]).ravel()

    def from_dev(a):
        o = np.empty((a.size // n, n))
        o[:, v2d] = a.reshape(-1, n)
        return o.ravel()

    prob = solvers.SolidBodyDrift(hp.SquareMeshP1(-1.0, 1.0, nc), Nt, dt, om=om, order=hp.ORDER_VERTEX)
    try:
        assert prob.ctx.kernel_regime(1) == hp._lib.REGIME_MESH
        ug = np.zeros_like(uo)
        ug[:n] = to_dev(u0)
        prob.solve_state(to_dev(ck), ug)
        pg = prob.solve_adjoint(to_dev(ck), ug, to_dev(uhat), np.zeros_like(ug), optim="alltime")
    finally:
        prob.close()
    eu, ep = rel(from_dev(ug), uo), rel(from_dev(pg), po)
    print(f"[mesh step] 41x41 vs oracle: u {eu:.2e}, p {ep:.2e}")
    assert eu < 1e-10 and ep < 1e-9
