"""CPU checks of the oracle's projected-gradient loop (oracle/pdeco.py): the bookkeeping the
refactored drivers share (descent, Armijo trial counts, failed line searches -> restore)."""
import numpy as np

from oracle import pdeco, traj as otraj
from oracle.assembly import P1Assembler
from oracle.mesh import SquareMesh


def _case(problem, Nt, dt):
    mesh = SquareMesh(0.0, 1.0, 8)
    asm = P1Assembler(mesh)
    n = mesh.nodes
    tl = (Nt + 1) * n
    rng = np.random.default_rng(3)
    z = lambda x0: np.concatenate([x0, np.zeros(Nt * n)])
    if problem == "nonlinear":
        u0 = np.exp(-20 * ((mesh.x - 0.5) ** 2 + (mesh.y - 0.5) ** 2))[mesh.dof_to_vertex]
        ut, _ = otraj.solve_nonlinear_equation(np.full(tl, 0.5), z(u0), None, asm, n, Nt, dt)
        return asm, (u0,), (ut[Nt * n:].copy(),)
    u0, v0 = 1 + 0.1 * rng.random(n), 0.9 + 0.1 * rng.random(n)
    ut, vt = otraj.solve_schnak_system(np.full(tl, 0.1), z(u0), z(v0), asm, n, Nt, dt)
    return asm, (u0, v0), (ut[Nt * n:].copy(), vt[Nt * n:].copy())


def test_nonlinear_loop_descends_and_respects_box():
    asm, ic, tg = _case("nonlinear", 6, 2e-3)
    r = pdeco.projected_gradient_descent("nonlinear", asm, asm.mass(), ic, tg, 6, 2e-3, max_iter_GD=3)
    assert r["it"] == 3 and not r["restored"]
    assert all(b < a for a, b in zip(r["cost"], r["cost"][1:]))
    assert r["c"].min() >= -1.0 and r["c"].max() <= 1.0
    assert all(1 <= k <= 5 for k in r["armijo_its"])


def test_failed_line_searches_end_the_loop_and_restore_the_control():
    asm, ic, tg = _case("schnak", 4, 1e-3)
    # one trial step of size 1 never satisfies the Armijo condition here: every line search "fails"
    r = pdeco.projected_gradient_descent("schnak", asm, asm.mass(), ic, tg, 4, 1e-3, max_iter_armijo=1, max_iter_GD=10)
    assert r["armijo_its"] == [1, 1]            # the third failure breaks before the metrics are appended
    assert r["restored"] and r["it"] == 2 and r["it_backup"] == 0
    assert len(r["cost"]) == 3


def test_oracle_schnak_time_dependent_wind_reduces_to_stationary():
    """oracle.traj: the separable wind s(t) w0(x) (Schnak_FCT_PDECO_alltime.py:55,174-175) with s == 1 is the HEAD
    solver; with s == 0 the convection drops out of both species (pure reaction-diffusion: the wind cannot matter)."""
    import numpy as np
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import traj as otraj
    asm = P1Assembler(SquareMesh(0, 1, 8))
    n, Nt, dt = asm.n, 4, 1e-3
    rng = np.random.default_rng(0)
    z = lambda: np.concatenate([1.0 + 0.1 * rng.random(n), np.zeros(Nt * n)])
    u0, v0 = z(), z()
    c = 0.1 * np.ones((Nt + 1) * n)
    a = otraj.solve_schnak_system(c, u0.copy(), v0.copy(), asm, n, Nt, dt)
    b = otraj.solve_schnak_system(c, u0.copy(), v0.copy(), asm, n, Nt, dt, wind=otraj.schnak_wind, wind_scale=lambda t: 1.0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    w1 = lambda x, y: (-(y - 0.5), (x - 0.5))
    w2 = lambda x, y: (x, 2 * y)
    p = otraj.solve_schnak_system(c, u0.copy(), v0.copy(), asm, n, Nt, dt, wind=w1, wind_scale=lambda t: 0.0)
    q = otraj.solve_schnak_system(c, u0.copy(), v0.copy(), asm, n, Nt, dt, wind=w2, wind_scale=lambda t: 0.0)
    assert np.allclose(p[0], q[0], rtol=0, atol=1e-14) and np.allclose(p[1], q[1], rtol=0, atol=1e-14)
    assert np.abs(a[0] - p[0]).max() > 1e-6


def test_solidbody_pgd_loop_records_margins_consistent_with_its_decisions():
    """oracle.traj.solidbody_pgd_loop (the inline Armijo loop of ..._finaltime_Garvie.py:259-317): a trial is rejected
    exactly when its recorded margin (J_trial - J_k + gam/s ||c_inc - c||^2_Q) / |J_k| is positive; the search stops at
    the first accepted trial or after max_armijo trials; both misfit variants."""
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle import traj as otraj
    mesh = SquareMesh(-1, 1, 8)
    asm = P1Assembler(mesh)
    n, Nt, dt = mesh.nodes, 4, 2e-3
    tl = (Nt + 1) * n
    u0 = np.exp(-15 * ((mesh.x + 0.2) ** 2 + (mesh.y - 0.1) ** 2))[mesh.dof_to_vertex]
    sb = otraj.SolidBody(asm)
    for optim in ("finaltime", "alltime"):
        if optim == "alltime":
            uhat = np.zeros(tl); uhat[:n] = u0
            otraj.solidbody_forward(sb, 2.0 * np.ones(tl), uhat, n, Nt, dt)
        else:
            uhat = np.exp(-15 * ((mesh.x + 0.1) ** 2 + (mesh.y - 0.2) ** 2))[mesh.dof_to_vertex]
        u, p, c, h = otraj.solidbody_pgd_loop(sb, u0, uhat, np.ones(tl), 0.05, 0.0, 5.0, 3, n, Nt, dt, max_armijo=4, optim=optim)
        assert u.shape == (tl,) and p.shape == (tl,) and c.shape == (tl,) and np.all((c >= 0.0) & (c <= 5.0))
        assert len(h["cost"]) == len(h["armijo_k"]) == len(h["armijo_margin"]) == 3
        for k, ms in zip(h["armijo_k"], h["armijo_margin"]):
            assert len(ms) == k and 1 <= k <= 4
            assert all(m > 0 for m in ms[:-1])                  # every trial before the last one looked at was rejected
            assert ms[-1] <= 0 or k == 4                        # the last one was accepted, or the trials ran out
        assert h["armijo_margin_min"] == min(abs(m) for ms in h["armijo_margin"] for m in ms)
