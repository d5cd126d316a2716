"""GPU parity tests of the three PDE systems' trajectory solvers and the BiCGStab species solve
(-m gpu): the drop-in ``solve_*`` functions against the CPU oracle and, for the chemotaxis forward
problem, directly against the real-FEniCS trajectory shipped by the reference."""
import importlib
import os

import numpy as np
import pytest
from scipy.sparse.linalg import spsolve

from helpers_golden import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    mod = importlib.import_module("fem-fct-pdeco_amd")
    mod.fct_helpers.VERBOSE = False
    return mod


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _oracle(a1, a2, nc):
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    mesh = SquareMesh(a1, a2, nc)
    return mesh, P1Assembler(mesh)


def test_chemotaxis_forward_vs_real_fenics_trajectory(hp):
    """Chtxs_data_dx0.025_dt0.001/chtxs_{m,f}_t0.01.csv: 10 steps of solve_chtxs_system computed by
    the reference with real FEniCS (control_fun=Constant(100), rescaling=1)."""
    z = load("chtxs_fenics_traj.npz")
    V = hp.SquareMeshP1(0.0, 1.0, 40)
    n, Nt, dt = V.nodes, 10, 1e-3
    m_ref = z["m"].reshape(Nt + 1, n)
    f_ref = z["f"].reshape(Nt + 1, n)
    u0, v0 = hp.chtxs_sys_IC(0, 1, 0.025, n, V.vertex_to_dof)
    assert np.array_equal(u0, m_ref[0]) and np.array_equal(v0, f_ref[0])
    u = np.zeros((Nt + 1) * n)
    v = np.zeros((Nt + 1) * n)
    u[:n], v[:n] = u0, v0
    ru, rv = hp.solve_chtxs_system(None, u, v, V, n, Nt, dt, None, control_fun=100, rescaling=1)
    assert ru is u and rv is v
    for k in range(1, Nt + 1):
        assert np.max(np.abs(v[k * n:(k + 1) * n] - f_ref[k])) < 1e-10, k
        assert np.max(np.abs(u[k * n:(k + 1) * n] - m_ref[k])) < 1e-10, k


def test_bicgstab_vs_spsolve(hp):
    from oracle.traj import schnak_wind
    mesh, asm = _oracle(0.0, 1.0, 40)
    n = mesh.nodes
    rng = np.random.default_rng(1)
    M, Ad = asm.mass(), asm.stiffness()
    A = asm.convection(schnak_wind)
    u = 1 + 0.2 * rng.random(n)
    mats = [M + 1e-3 * (8.6676 * Ad - 0.6 * A + 230.82 * asm.weighted_mass(lambda at: at(u) ** 2)),   # helpers.py:595
            M + 1e-3 * (0.05 * Ad + 100 * M)]                                                            # helpers.py:1308
    ctx = hp.Context(0)
    Mc = M.copy()
    Mc.sort_indices()
    ctx.set_pattern_csr(Mc.indptr, Mc.indices)
    ctx.set_mass(Mc.data, np.asarray(M.sum(axis=1)).ravel())
    for Mat in mats:
        Mat = Mat.tocsr()
        Mat.sort_indices()
        b = rng.standard_normal(n)
        ell = ctx.csr_to_ell(Mat.data)
        x = ctx.empty(n)
        info = ctx.bicgstab(ell, ctx.array(b), ctx.array(np.zeros(n)), x)
        xs = spsolve(Mat.tocsc(), b)
        assert rel(x.download(), xs) < 1e-10, info
        assert info[0]["solver_resid"] <= 1e-13
    ctx.close()


def test_nonlinear_forward_adjoint_vs_oracle(hp):
    from oracle import traj as otraj
    mesh, asm = _oracle(0.0, 1.0, 20)
    V = hp.SquareMeshP1(0.0, 1.0, 20)
    n, Nt, dt = V.nodes, 8, 5e-3
    rng = np.random.default_rng(2)
    u0 = hp.nonlinear_equation_IC(0, 1, 0.05, n, V.vertex_to_dof)
    ctrl = rng.random((Nt + 1) * n)
    uo = np.zeros((Nt + 1) * n); uo[:n] = u0
    ug = uo.copy()
    otraj.solve_nonlinear_equation(ctrl, uo, None, asm, n, Nt, dt)
    r = hp.solve_nonlinear_equation(ctrl, ug, None, V, n, Nt, dt, None)
    assert r[0] is ug and r[1] is None
    assert rel(ug, uo) < 1e-9
    # control_fun (constant) path
    uo2 = np.zeros((Nt + 1) * n); uo2[:n] = u0
    ug2 = uo2.copy()
    otraj.solve_nonlinear_equation(None, uo2, None, asm, n, Nt, dt, control_const=0.7)
    hp.solve_nonlinear_equation(None, ug2, None, V, n, Nt, dt, None, control_fun=0.7)
    assert rel(ug2, uo2) < 1e-9
    uhat = 0.8 * uo[Nt * n:] + 0.01
    po = otraj.solve_adjoint_nonlinear_equation(uo, uhat, np.zeros_like(uo), Nt * dt, asm, n, Nt, dt)
    pg = hp.solve_adjoint_nonlinear_equation(ug, uhat, np.zeros_like(ug), Nt * dt, V, n, Nt, dt, None)
    assert rel(pg, po) < 1e-9


def test_schnak_forward_adjoint_vs_oracle(hp):
    from oracle import traj as otraj
    mesh, asm = _oracle(0.0, 1.0, 20)
    V = hp.SquareMeshP1(0.0, 1.0, 20)
    n, Nt, dt = V.nodes, 6, 1e-3
    rng = np.random.default_rng(3)
    u0, v0 = hp.schnak_sys_IC(0, 1, 0.05, n, V.vertex_to_dof)
    ctrl = 0.1 + 0.05 * rng.random((Nt + 1) * n)
    uo = np.zeros((Nt + 1) * n); vo = np.zeros((Nt + 1) * n)
    uo[:n], vo[:n] = u0, v0
    ug, vg = uo.copy(), vo.copy()
    otraj.solve_schnak_system(ctrl, uo, vo, asm, n, Nt, dt)
    hp.solve_schnak_system(ctrl, ug, vg, V, n, Nt, dt, None)
    assert rel(ug, uo) < 1e-9 and rel(vg, vo) < 1e-9
    uhat, vhat = 0.9 * uo[Nt * n:], 1.1 * vo[Nt * n:]
    po, qo = otraj.solve_adjoint_schnak_system(uo, vo, uhat, vhat, np.zeros_like(uo), np.zeros_like(uo), Nt * dt, asm, n, Nt, dt)
    pg, qg = hp.solve_adjoint_schnak_system(ug, vg, uhat, vhat, np.zeros_like(ug), np.zeros_like(ug), Nt * dt, V, n, Nt, dt, None)
    assert rel(pg, po) < 1e-8 and rel(qg, qo) < 1e-8
    # all-time misfit (extension; config C3): trajectory targets, zero terminal conditions
    uh, vh = 0.9 * uo + 0.01, 1.1 * vo
    po, qo = otraj.solve_adjoint_schnak_system(uo, vo, uh, vh, np.zeros_like(uo), np.zeros_like(uo), Nt * dt, asm, n, Nt, dt,
                                               None, "alltime")
    pg, qg = hp.solve_adjoint_schnak_system(ug, vg, uh, vh, np.ones_like(ug), np.ones_like(ug), Nt * dt, V, n, Nt, dt, None,
                                            optim="alltime")
    assert rel(pg, po) < 1e-8 and rel(qg, qo) < 1e-8
    assert not pg[Nt * n:].any() and not qg[Nt * n:].any()
    with pytest.raises(ValueError):
        hp.solve_adjoint_schnak_system(ug, vg, uh, vh, pg, qg, Nt * dt, V, n, Nt, dt, None, optim="sometime")


@pytest.mark.parametrize("optim", ["alltime", "finaltime"])
def test_chemotaxis_forward_adjoint_vs_oracle(hp, optim):
    from oracle import traj as otraj
    mesh, asm = _oracle(0.0, 1.0, 20)
    V = hp.SquareMeshP1(0.0, 1.0, 20)
    n, Nt, dt = V.nodes, 6, 5e-4
    rng = np.random.default_rng(4)
    u0 = 1.5 + 0.1 * (0.5 - rng.random(n))
    ctrl = 20 * rng.random((Nt + 1) * n)
    uo = np.zeros((Nt + 1) * n); vo = np.zeros((Nt + 1) * n)
    uo[:n], vo[:n] = u0, u0
    ug, vg = uo.copy(), vo.copy()
    otraj.solve_chtxs_system(ctrl, uo, vo, asm, n, Nt, dt)
    hp.solve_chtxs_system(ctrl, ug, vg, V, n, Nt, dt, None)
    assert rel(ug, uo) < 1e-9 and rel(vg, vo) < 1e-9
    if optim == "alltime":
        uhat, vhat = 0.9 * uo + 0.01 * rng.random(uo.size), 1.05 * vo
    else:
        uhat, vhat = 0.9 * uo[Nt * n:], 1.05 * vo[Nt * n:]
    po, qo = otraj.solve_adjoint_chtxs_system(uo, vo, uhat, vhat, np.zeros_like(uo), np.zeros_like(uo), ctrl, Nt * dt,
                                              asm, n, Nt, dt, None, optim)
    pg, qg = hp.solve_adjoint_chtxs_system(ug, vg, uhat, vhat, np.zeros_like(ug), np.zeros_like(ug), ctrl, Nt * dt,
                                           V, n, Nt, dt, None, optim)
    assert rel(pg, po) < 1e-8 and rel(qg, qo) < 1e-8
    with pytest.raises(ValueError):
        hp.solve_adjoint_chtxs_system(ug, vg, uhat, vhat, pg, qg, ctrl, Nt * dt, V, n, Nt, dt, None, "sometime")


def test_species_chebyshev_matches_bicgstab(hp):
    """The tile-fused Chebyshev species solve (vertex order) and BiCGStab meet the same tolerance:
    identical Schnakenberg / chemotaxis trajectories to solver accuracy; the flag tells which ran."""
    import os
    systems = importlib.import_module("fem-fct-pdeco_amd.systems")
    V = hp.SquareMeshP1(0.0, 1.0, 40)
    n, Nt, dt = V.nodes, 12, 5e-4
    rng = np.random.default_rng(11)
    # the Chebyshev variant needs the tile kernels (tuning knobs may switch them off: then BiCGStab runs)
    tiles_on = all(os.environ.get(k, "1") != "0" for k in ("FEMFCT_TILES", "FEMFCT_STRIPS", "FEMFCT_IMPLICIT"))
    S = systems.PDESystems(V, order=hp.ORDER_VERTEX)
    ctx = S.ctx
    try:
        par, wind = systems._schnak_par()
        Aw, AwT = S.convection(wind, "schnak")
        u0, v0 = hp.schnak_sys_IC(0, 1, 0.025, n, np.arange(n))
        c = ctx.array(0.1 + 0.01 * rng.random(n))
        cpar = systems._chtxs_par()
        cc = ctx.array(20 * rng.random(n))
        uc0 = 1.5 + 0.1 * (0.5 - rng.random(n))
        uh, vh = rng.random(n), rng.random(n)
        res = {}
        for mode in ("auto", "bicgstab"):
            ctx.set_species_solver(mode)
            def traj(x0):
                a = np.zeros((Nt + 1) * n)
                a[:n] = x0
                return ctx.array(a)
            u, v, p, q = traj(u0), traj(v0), traj(0 * u0), traj(0 * u0)
            ctx.schnak_forward(Aw, c, u, v, Nt, dt, par, 1.0)
            kf = ctx.traj_krylov_info(Nt)
            ctx.schnak_adjoint(AwT, u, v, ctx.array(uh), ctx.array(vh), p, q, Nt, dt, par)
            ka = ctx.traj_krylov_info(Nt)
            uc, vc = traj(uc0), traj(uc0)
            ctx.chtxs_forward(cc, uc, vc, Nt, dt, cpar, 0.1)
            kc = ctx.traj_krylov_info(Nt)
            res[mode] = [x.download() for x in (u, v, p, q, uc, vc)]
            for k in (kf, ka, kc):
                assert np.all((k["flags"] & hp.FLAG_SOLVER_BUDGET) == 0)
                if tiles_on:
                    assert np.all(((k["flags"] & hp.FLAG_CHEBYSHEV) != 0) == (mode == "auto"))
                assert k["solver_resid"].max() <= 1e-13
        for a, b in zip(res["auto"], res["bicgstab"]):
            assert rel(a, b) < 1e-10
    finally:
        S.close()


@pytest.mark.parametrize("batch", [1, 3])
def test_form_groups_and_fused_step_end_are_bitwise_neutral(hp, monkeypatch, batch):
    """Two launch-count savers of the systems' time steps must not change a bit or a log entry: (i) FEMFCT_FORM_GROUPS --
    the quadrature forms of a step that depend on earlier levels only share one launch (k_forms2 / k_forms3), and the
    chemotaxis q right-hand side M q_{n+1} + dt rhs_q is formed in the pass that evaluates rhs_q; (ii) FEMFCT_FUSE_END --
    the one-launch species solve logs the step and moves the time level itself when it is the step's last operation.
    Schnakenberg forward + all-time adjoint, chemotaxis forward + all-time adjoint, 60 steps (two graphs of 50 / 10 steps),
    states, adjoints, FCT-step log and species-solve log."""
    systems = importlib.import_module("fem-fct-pdeco_amd.systems")
    V = hp.SquareMeshP1(0.0, 1.0, 40)
    n, Nt, dt = V.nodes, 60, 5e-4
    tl = (Nt + 1) * n
    outs = []
    for groups, fuse_end in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("FEMFCT_FORM_GROUPS", groups)
        monkeypatch.setenv("FEMFCT_FUSE_END", fuse_end)
        rng = np.random.default_rng(23)
        S = systems.PDESystems(V, order=hp.ORDER_VERTEX)
        ctx = S.ctx
        try:
            def traj(x0):
                a = np.zeros((batch, tl))
                a[:, :n] = x0
                return ctx.array(a.ravel())
            par, wind = systems._schnak_par()
            Aw, AwT = S.convection(wind, "schnak")
            u0, v0 = hp.schnak_sys_IC(0, 1, 0.025, n, np.arange(n))
            c = ctx.array(0.1 + 0.01 * rng.random((batch, n)).ravel())
            u, v, p, q = traj(u0), traj(v0), traj(0 * u0), traj(0 * u0)
            ctx.schnak_forward(Aw, c, u, v, Nt, dt, par, 1.0, batch=batch)
            logs = [ctx.traj_info(Nt, batch), ctx.traj_krylov_info(Nt, batch)]
            uh, vh = ctx.array(rng.random(batch * tl)), ctx.array(rng.random(batch * tl))
            ctx.schnak_adjoint(AwT, u, v, uh, vh, p, q, Nt, dt, par, batch=batch, alltime=True)
            logs += [ctx.traj_info(Nt, batch), ctx.traj_krylov_info(Nt, batch)]
            cpar = systems._chtxs_par()
            cc = ctx.array(20 * rng.random(batch * tl))
            uc0 = 1.5 + 0.1 * (0.5 - rng.random(n))
            uc, vc, pc, qc = traj(uc0), traj(uc0), traj(0 * uc0), traj(0 * uc0)
            ctx.chtxs_forward(cc, uc, vc, Nt, dt, cpar, 0.1, batch=batch)
            logs += [ctx.traj_info(Nt, batch), ctx.traj_krylov_info(Nt, batch)]
            ctx.chtxs_adjoint(uc, vc, uh, vh, pc, qc, cc, Nt, dt, cpar, 0.1, alltime=True, batch=batch)
            logs += [ctx.traj_info(Nt, batch), ctx.traj_krylov_info(Nt, batch)]
            outs.append(([x.download() for x in (u, v, p, q, uc, vc, pc, qc)], logs))
        finally:
            S.close()
    for arrs, logs in outs[1:]:
        for a, b in zip(arrs, outs[0][0]):
            assert np.isfinite(a).all() and np.array_equal(a, b)
        for la, lb in zip(logs, outs[0][1]):
            for k in ("flags", "solver_iters", "solver_resid"):
                assert np.array_equal(la[k], lb[k]), k
    assert outs[0][0][2].any() and outs[0][0][6].any()


def test_mimura_named_config_forward_synthetic(hp, monkeypatch):
    """The 'Mimura-Tsujikawa' scripts at HEAD (chemotaxis_mimura_FCT.py:25-44, mimura_data_helpers.py:82-100) run
    the same chemotaxis operators with delta = 2, Dm = Df = 0.05, chi = 0.125, beta = 0.5 on [0,10]^2: the device
    sweep takes the parameters and the domain as inputs -- forward synthetic against the oracle."""
    from oracle import traj as otraj
    systems = importlib.import_module("fem-fct-pdeco_amd.systems")
    monkeypatch.setattr(otraj, "chtxs_params",
                        lambda: dict(delta=2, Dm=0.05, Df=0.05, chi=0.125, gamma=100, eta=0.5))
    mesh, asm = _oracle(0.0, 10.0, 24)
    V = hp.SquareMeshP1(0.0, 10.0, 24)
    n, Nt, dt = V.nodes, 8, 0.01
    rng = np.random.default_rng(9)
    m0 = 1.0 + 0.05 * rng.random(n)
    uo = np.zeros((Nt + 1) * n); vo = np.zeros((Nt + 1) * n)
    uo[:n], vo[:n] = m0, m0 / 2
    otraj.solve_chtxs_system(None, uo, vo, asm, n, Nt, dt, control_const=1.0, rescaling=1)
    S = systems.PDESystems(V, order=hp.ORDER_FENICS)
    try:
        ctx = S.ctx
        u, v = ctx.array(np.concatenate([m0, np.zeros(Nt * n)])), ctx.array(np.concatenate([m0 / 2, np.zeros(Nt * n)]))
        ctx.chtxs_forward(ctx.array(np.full(n, 1.0)), u, v, Nt, dt, [2, 0.05, 0.05, 0.125, 0.5], 1.0)
        ug, vg = u.download(), v.download()
    finally:
        S.close()
    assert rel(ug, uo) < 1e-9 and rel(vg, vo) < 1e-9


def test_assemble_mass_and_armijo_line_search_ref_match_oracle(hp):
    """assemble_mass (FEniCS DoF order) and the drop-in armijo_line_search_ref (helpers.py:1583-1713) against the
    oracle's restatement on the nonlinear problem: same accepted trial, control, state and clobbering behaviour."""
    from oracle import fct as ofct, traj as otraj
    systems = importlib.import_module("fem-fct-pdeco_amd.systems")
    mesh, asm = _oracle(0.0, 1.0, 12)
    V = hp.SquareMeshP1(0.0, 1.0, 12)
    n, Nt, dt = V.nodes, 8, 2e-3
    M = hp.assemble_mass(V)
    Mo = asm.mass().tocsr()
    assert abs(M - Mo).max() < 1e-16 and M.nnz == Mo.nnz
    # convection matrix of the nonlinear wind through the same route
    eps, _, wind = hp.get_nonlinear_eqns_params()
    Aw = systems.device_matrix(V, systems._system(V).convection(wind, "nonlinear")[0])
    assert abs(Aw - asm.convection(otraj.nonlinear_wind)).max() < 1e-15
    tl = (Nt + 1) * n
    rng = np.random.default_rng(5)
    u0 = hp.nonlinear_equation_IC(0, 1, 1 / 12, n, V.vertex_to_dof)
    z = lambda: np.concatenate([u0, np.zeros(Nt * n)])
    c = 0.2 * np.ones(tl)
    uo, _ = otraj.solve_nonlinear_equation(c, z(), None, asm, n, Nt, dt)
    target = 0.8 * uo[Nt * n:] + 0.01
    d = 0.5 * rng.standard_normal(tl)
    J0 = ofct.cost_functional(uo, target, c, Nt, dt, Mo, 0.1, "finaltime")
    solve_o = lambda ci, v1, v2: otraj.solve_nonlinear_equation(ci, v1, v2, asm, n, Nt, dt)
    ro = ofct.armijo_line_search(uo.copy(), c, d, target, Nt, dt, -1.0, 1.0, 0.1, J0, n, "finaltime", Mo, max_iter=5,
                                 nonlinear_solver=solve_o)
    rg = hp.armijo_line_search_ref(uo.copy(), c, d, target, Nt, dt, -1.0, 1.0, 0.1, J0, n, "finaltime", V, max_iter=5,
                                   nonlinear_solver=hp.solve_nonlinear_equation)
    assert rg[2] == ro[2]                                   # k + 1
    assert rel(rg[1], ro[1]) < 1e-14 and rel(rg[0], ro[0]) < 1e-9
    with pytest.raises(ValueError):
        hp.armijo_line_search_ref(uo, c, d, target, Nt, dt, -1.0, 1.0, 0.1, J0, n, "sometime", V,
                                  nonlinear_solver=hp.solve_nonlinear_equation)


def test_chemotaxis_adjoint_only_harness(hp):
    """The reference's solve-the-adjoint-only harness (chemotaxis_adjoint_equations.py:94-107): constant states
    uhat = 1, vhat = 2, u = 0.8 uhat, v = 0.8 vhat, control 100, all-time misfit -- device sweep against the oracle,
    plus what the constant data implies: spatially constant adjoints (every node sees the same equations)."""
    from oracle import traj as otraj
    mesh, asm = _oracle(0.0, 1.0, 16)
    V = hp.SquareMeshP1(0.0, 1.0, 16)
    n, Nt, dt = V.nodes, 10, 1e-3
    tl = (Nt + 1) * n
    uhat, vhat = np.ones(tl), 2 * np.ones(tl)
    uk, vk, control = 0.8 * uhat, 0.8 * vhat, 100 * np.ones(tl)
    po, qo = otraj.solve_adjoint_chtxs_system(uk, vk, uhat, vhat, np.zeros(tl), np.zeros(tl), control, Nt * dt, asm, n, Nt,
                                              dt, None, "alltime")
    pg, qg = hp.solve_adjoint_chtxs_system(uk, vk, uhat, vhat, np.zeros(tl), np.zeros(tl), control, Nt * dt, V, n, Nt, dt,
                                           None, "alltime")
    assert rel(pg, po) < 1e-9 and rel(qg, qo) < 1e-9
    assert not pg[Nt * n:].any() and not qg[Nt * n:].any()          # p(T) = q(T) = 0
    assert np.abs(pg).max() > 0 and np.abs(qg).max() > 0


def test_species_solver_falls_back_to_bicgstab_when_chebyshev_runs_out_of_iterations(hp):
    """With an iteration cap between what BiCGStab (~70) and the Chebyshev iteration (~150) need for the
    Schnakenberg species solve, the sweep must notice the failed Chebyshev solves, switch that kind of sweep to
    BiCGStab, repeat it and still deliver the reference result; with a cap below both it must raise."""
    from oracle import traj as otraj
    systems = importlib.import_module("fem-fct-pdeco_amd.systems")
    mesh, asm = _oracle(0.0, 1.0, 40)
    V = hp.SquareMeshP1(0.0, 1.0, 40)
    n, Nt, dt = V.nodes, 4, 5e-4
    u0, v0 = hp.schnak_sys_IC(0, 1, 0.025, n, V.vertex_to_dof)
    z = lambda x0: np.concatenate([x0, np.zeros(Nt * n)])
    uo, vo = otraj.solve_schnak_system(np.full((Nt + 1) * n, 0.1), z(u0), z(v0), asm, n, Nt, dt)
    S = systems.PDESystems(V, order=hp.ORDER_FENICS)
    ctx = S.ctx
    try:
        par, wind = systems._schnak_par()
        Aw, _ = S.convection(wind, "schnak")
        c = ctx.array(np.full(n, 0.1))
        ctx.set_krylov(1e-13, 100)
        u, v = ctx.array(z(u0)), ctx.array(z(v0))
        ctx.schnak_forward(Aw, c, u, v, Nt, dt, par, 1.0)
        k = ctx.traj_krylov_info(Nt)
        assert np.all((k["flags"] & hp.FLAG_CHEBYSHEV) == 0) and np.all((k["flags"] & hp.FLAG_SOLVER_BUDGET) == 0)
        assert rel(u.download(), uo) < 1e-9 and rel(v.download(), vo) < 1e-9
        ctx.set_krylov(1e-13, 20)
        with pytest.raises(hp.NotConverged):
            ctx.schnak_forward(Aw, c, u, v, Nt, dt, par, 1.0)
    finally:
        S.close()


def test_schnak_time_dependent_wind_of_the_named_config3_script(hp):
    """Schnak_FCT_PDECO_alltime.py:55,174-175,228-230 (the script BASELINE config 3 names): the wind is
    (-(y-.5), (x-.5)) * sin(2 pi t), re-assembled every step.  Separable => A(t_n) = sin(2 pi t_n) * A0 with a
    per-level factor refreshed inside the captured graph.  Forward + final-time and all-time adjoints against the
    oracle at the script's dt = 2e-3 on a 13 x 13 mesh; a factor of one reproduces the stationary sweep bit for bit."""
    from oracle import traj as otraj
    mesh, asm = _oracle(0.0, 1.0, 12)
    V = hp.SquareMeshP1(0.0, 1.0, 12)
    n, Nt, dt = V.nodes, 12, 2e-3
    rot = lambda x, y: (-(y - 0.5), (x - 0.5))
    s_t = lambda t: np.sin(2 * np.pi * 40 * t)          # 40x faster than the script's so that 12 steps see a sign change
    rng = np.random.default_rng(6)
    u0, v0 = hp.schnak_sys_IC(0, 1, 1 / 12, n, V.vertex_to_dof)
    ctrl = 0.1 + 0.05 * rng.random((Nt + 1) * n)
    z = lambda x0: np.concatenate([x0, np.zeros(Nt * n)])
    uo, vo = otraj.solve_schnak_system(ctrl, z(u0), z(v0), asm, n, Nt, dt, wind=rot, wind_scale=s_t)
    ug, vg = hp.solve_schnak_system(ctrl, z(u0), z(v0), V, n, Nt, dt, None, wind=rot, wind_scale=s_t)
    assert rel(ug, uo) < 1e-9 and rel(vg, vo) < 1e-9
    # it is not the stationary problem
    us, _ = hp.solve_schnak_system(ctrl, z(u0), z(v0), V, n, Nt, dt, None, wind=rot)
    assert rel(us, ug) > 1e-4
    zz = lambda: np.zeros_like(uo)
    uhat, vhat = 0.9 * uo[Nt * n:], 1.1 * vo[Nt * n:]
    po, qo = otraj.solve_adjoint_schnak_system(uo, vo, uhat, vhat, zz(), zz(), Nt * dt, asm, n, Nt, dt, wind=rot, wind_scale=s_t)
    pg, qg = hp.solve_adjoint_schnak_system(ug, vg, uhat, vhat, zz(), zz(), Nt * dt, V, n, Nt, dt, None, wind=rot, wind_scale=s_t)
    assert rel(pg, po) < 1e-8 and rel(qg, qo) < 1e-8
    uh, vh = 0.9 * uo + 0.01, 1.1 * vo
    po, qo = otraj.solve_adjoint_schnak_system(uo, vo, uh, vh, zz(), zz(), Nt * dt, asm, n, Nt, dt, None, "alltime",
                                               wind=rot, wind_scale=s_t)
    pg, qg = hp.solve_adjoint_schnak_system(ug, vg, uh, vh, zz(), zz(), Nt * dt, V, n, Nt, dt, None, optim="alltime",
                                            wind=rot, wind_scale=s_t)
    assert rel(pg, po) < 1e-8 and rel(qg, qo) < 1e-8
    # factor one == the stationary path, bitwise
    u1, v1 = hp.solve_schnak_system(ctrl, z(u0), z(v0), V, n, Nt, dt, None, wind=rot, wind_scale=np.ones(Nt + 1))
    us, vs = hp.solve_schnak_system(ctrl, z(u0), z(v0), V, n, Nt, dt, None, wind=rot)
    assert np.array_equal(u1, us) and np.array_equal(v1, vs)
    with pytest.raises(ValueError):
        hp.solve_schnak_system(ctrl, z(u0), z(v0), V, n, Nt, dt, None, wind=rot, wind_scale=np.ones(Nt))


def test_unnamed_wind_cache_is_bounded(hp):
    """PDESystems.convection keys a bare wind function on the function object; a caller that builds a new lambda per call
    must not grow the device-side cache without bound."""
    systems = importlib.import_module("fem-fct-pdeco_amd.systems")
    S = systems.PDESystems(hp.SquareMeshP1(0.0, 1.0, 8))
    try:
        first = None
        for k in range(3 * S.MAX_UNNAMED_WINDS):
            A, AT = S.convection(lambda x, y, k=k: ((y - 0.5) * (1 + k), -(x - 0.5)))
            first = first or (A, AT)
        assert sum(1 for key in S._conv if callable(key)) <= S.MAX_UNNAMED_WINDS
        assert first[0].ptr == 0 and first[1].ptr == 0          # the oldest pair was freed
        named = S.convection(lambda x, y: (x, y), "mine")
        assert S.convection(lambda x, y: (y, x), "mine") is named       # named winds are kept and reused
    finally:
        S.close()


def test_graph_replay_is_active_outside_a_profiler(hp):
    ctx = hp.Context(0)
    try:
        if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCP_TOOL_LIBRARIES"):
            assert not ctx.graph_replay_active()
        else:
            assert ctx.graph_replay_active()
        ctx.set_graphs(False)
        assert not ctx.graph_replay_active()
    finally:
        ctx.close()
