"""Known-answer test of the oracle (CPU): the manufactured solution of advection_FCT_PDECO_alltime_exact.py
(config C1's parameter set).  With the exact control the FCT state and adjoint sweeps converge to the
closed-form u_ex, p_ex at second order -- independent evidence for the sign / time-level conventions."""
import numpy as np

from oracle import traj as otraj
from oracle.assembly import P1Assembler
from oracle.mesh import SquareMesh


def _run(nc):
    mesh = SquareMesh(0.0, 1.0, nc)
    asm = P1Assembler(mesh)
    dx = 1.0 / nc
    dt = dx ** 2
    Nt = round(1.0 / dt)
    n = mesh.nodes
    g = np.arange(0.0, 1.0 + dx, dx)[:nc + 1]
    X, Y = np.meshgrid(g, g)
    f = [otraj.exact_fields(i * dt, X, Y) for i in range(Nt + 1)]
    st = lambda k: np.concatenate([fi[k].reshape(n)[mesh.dof_to_vertex] for fi in f])
    F = {k: st(k) for k in ("u", "p", "c", "g", "uhat")}
    ls = otraj.LinearSource(asm, eps=1e-3)
    u = np.zeros((Nt + 1) * n)
    u[:n] = F["u"][:n]
    otraj.linear_forward(ls, F["g"] + F["c"], u, n, Nt, dt)
    p = otraj.linear_adjoint(ls, u, F["uhat"], np.zeros_like(u), n, Nt, dt)
    r = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    return r(u, F["u"]), r(p, F["p"])


def test_manufactured_solution_second_order():
    eu10, ep10 = _run(10)
    eu20, ep20 = _run(20)
    assert eu10 < 0.02 and ep10 < 0.05
    assert eu20 < 0.3 * eu10 and ep20 < 0.3 * ep10
