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
