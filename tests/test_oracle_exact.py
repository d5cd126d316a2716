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


def test_rotation_operator_closed_form_equals_the_quadrature():
    """What femfct_assemble_rotation / sb_rot_row (csrc/solidbody_op.h) evaluate: for the linear wind w = omega (-y, x)
    the element matrix of dot(w, grad(v))*u*dx is grad(lambda_i) . |K|/12 (w_0 + w_1 + w_2 + w_j) -- restated here with
    the oracle's element data and compared with the oracle's own quadrature assembly of the same form."""
    import numpy as np
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler
    from oracle.traj import rotation_wind
    for a1, a2, nc, om in ((-1.0, 1.0, 12, np.pi / 40), (0.0, 1.0, 7, 0.3)):
        asm = P1Assembler(SquareMesh(a1, a2, nc))
        wx, wy = rotation_wind(om)(asm.xv[..., 0], asm.xv[..., 1])            # wind at the three vertices of every cell
        w = np.stack([wx, wy], axis=-1)                                       # (nt, 3, 2)
        tot = w.sum(axis=1, keepdims=True) + w                                # W + w_j
        Ke = (asm.area[:, None, None] / 12.0) * np.einsum("tid,tjd->tij", asm.grad, tot)
        A_closed = asm._mat(Ke)
        A_quad = asm.convection(rotation_wind(om))
        assert abs(A_closed - A_quad).max() < 1e-14 * max(1.0, abs(A_quad).max())
