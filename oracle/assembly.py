"""P1 finite-element assembly on the structured mesh (what the reference gets
from ``df.assemble`` through helpers.py:87-141).

TEST INFRASTRUCTURE (see oracle/__init__.py).

dolfin is a third-party dependency that is absent from /root/reference and
from this image (no pin file in the reference; 2019-era FEniCS).  Its
published algorithm for these forms is: per-cell Gauss quadrature of the UFL
integrand with the degree UFL estimates (sum of polynomial degrees; exp(f)
counts as degree(f)+2), "default" FIAT triangle scheme.  All polynomial forms
are therefore integrated exactly and any exact rule agrees to rounding; the
only rule-dependent forms are the chemotaxis exp-forms:
  * forward  (helpers.py:1350-1351)  estimated degree 4 -> 6-point rule
    [pinned by the shipped FEniCS trajectory],
  * adjoint matrix (helpers.py:1499-1500) estimated degree 5 -> 7-point rule
    [inferred, unpinned],
  * adjoint rhs (helpers.py:1531-1532) estimated degree 4 -> 6-point rule
    [inferred, unpinned].

Row index = test function, column index = trial function, everything returned
in FEniCS DoF order as ``scipy.sparse.csr_matrix`` / 1-D float64 arrays.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import coo_matrix, csr_matrix

# ---------------------------------------------------------------------------
# Quadrature rules on the reference triangle, barycentric (l0,l1,l2), weights
# normalised to sum to 1 (multiply by |K|).
# ---------------------------------------------------------------------------
_a = 0.091576213509771
_b = 0.445948490915965
_wa = 0.109951743655322
_wb = 0.223381589678011
QUAD6_PTS = np.array([
    [1 - 2 * _a, _a, _a], [_a, 1 - 2 * _a, _a], [_a, _a, 1 - 2 * _a],
    [1 - 2 * _b, _b, _b], [_b, 1 - 2 * _b, _b], [_b, _b, 1 - 2 * _b]])
QUAD6_W = np.array([_wa, _wa, _wa, _wb, _wb, _wb])

_c = 0.10128650732345633
_d = 0.47014206410511505
QUAD7_PTS = np.array([
    [1 / 3, 1 / 3, 1 / 3],
    [1 - 2 * _c, _c, _c], [_c, 1 - 2 * _c, _c], [_c, _c, 1 - 2 * _c],
    [1 - 2 * _d, _d, _d], [_d, 1 - 2 * _d, _d], [_d, _d, 1 - 2 * _d]])
QUAD7_W = np.array([0.225, 0.12593918054482717, 0.12593918054482717,
                    0.12593918054482717, 0.13239415278850616,
                    0.13239415278850616, 0.13239415278850616])


class P1Assembler:
    """Per-mesh geometry cache + scatter maps."""

    def __init__(self, mesh):
        self.mesh = mesh
        c = mesh.cells
        self.nt = c.shape[0]
        x = np.stack([mesh.x[c], mesh.y[c]], axis=2)  # (nt,3,2)
        e1 = x[:, 1] - x[:, 0]
        e2 = x[:, 2] - x[:, 0]
        det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
        self.area = 0.5 * np.abs(det)
        # gradients of barycentric coordinates (constant per cell)
        g = np.empty((self.nt, 3, 2))
        g[:, 1, 0] = e2[:, 1] / det
        g[:, 1, 1] = -e2[:, 0] / det
        g[:, 2, 0] = -e1[:, 1] / det
        g[:, 2, 1] = e1[:, 0] / det
        g[:, 0] = -(g[:, 1] + g[:, 2])
        self.grad = g
        self.xv = x
        dof = mesh.vertex_to_dof[c]  # (nt,3) global DoF per local vertex
        self.dof = dof
        self.rows = np.repeat(dof, 3, axis=1).reshape(-1)       # i slow
        self.cols = np.tile(dof, (1, 3)).reshape(-1)            # j fast
        n = mesh.nodes
        self.n = n
        self.MK = (self.area[:, None, None] / 12.0) * (np.ones((3, 3)) + np.eye(3))[None]

    # -- scatter ---------------------------------------------------------
    def _mat(self, Ke) -> csr_matrix:
        A = coo_matrix((Ke.reshape(-1), (self.rows, self.cols)), shape=(self.n, self.n)).tocsr()
        A.sort_indices()
        return A

    def _vec(self, be) -> np.ndarray:
        out = np.zeros(self.n)
        np.add.at(out, self.dof.reshape(-1), be.reshape(-1))
        return out

    def _local(self, vec_dof):
        """P1 function given in DoF order -> (nt,3) local vertex values."""
        return np.asarray(vec_dof, dtype=np.float64)[self.dof]

    def _at(self, loc, pts):
        """(nt,3) vertex values -> (nt,nq) values at barycentric pts."""
        return loc @ pts.T

    # -- constant forms --------------------------------------------------
    def mass(self):
        """u*v*dx (helpers.py:553,655,930,1012,1305,1470)."""
        return self._mat(self.MK)

    def stiffness(self):
        """dot(grad(u),grad(v))*dx (helpers.py:555,657,932,1014,1307,1472)."""
        g = self.grad
        Ke = self.area[:, None, None] * np.einsum("tid,tjd->tij", g, g)
        return self._mat(Ke)

    def convection(self, wind, pts=QUAD7_PTS, w=QUAD7_W):
        """dot(wind, grad(v))*u*dx: A[i,j] = int (w . grad phi_i) phi_j
        (helpers.py:581,933,1015; advection_solidbody_FCT_PDECO_finaltime.py:122).
        ``wind(x,y) -> (wx,wy)``, polynomial of degree <= 4 in the reference
        (``df.Expression(..., degree=4)``) so a degree-5 rule is exact."""
        xq = np.einsum("qa,tad->tqd", pts, self.xv)  # (nt,nq,2)
        wx, wy = wind(xq[..., 0], xq[..., 1])
        wg = wx[:, :, None] * self.grad[:, None, :, 0] + wy[:, :, None] * self.grad[:, None, :, 1]  # (nt,nq,3) = w.grad phi_i
        Ke = np.einsum("t,q,tqi,qj->tij", self.area, w, wg, pts)
        return self._mat(Ke)

    # -- forms with P1 coefficient functions -----------------------------
    def weighted_mass(self, fq_fn, pts=QUAD7_PTS, w=QUAD7_W):
        """int f phi_i phi_j with f given at quadrature points by
        ``fq_fn(at)`` where ``at(vec_dof)`` evaluates a P1 function there.
        Covers M_u2 = u_h^2*u*v*dx, M_uv = u_h*v_h*u*v*dx
        (helpers.py:591,683,692,953,1032)."""
        fq = fq_fn(lambda vec: self._at(self._local(vec), pts))
        Ke = np.einsum("t,q,tq,qi,qj->tij", self.area, w, fq, pts, pts)
        return self._mat(Ke)

    def load(self, fq_fn, pts=QUAD7_PTS, w=QUAD7_W):
        """int f phi_i (helpers.py:584-585,594,684,693,956,1339-1340,1505)."""
        fq = fq_fn(lambda vec: self._at(self._local(vec), pts))
        be = np.einsum("t,q,tq,qi->ti", self.area, w, fq, pts)
        return self._vec(be)

    def drift1(self, c_dof, b=(1.0, 1.0)):
        """dot(drift, grad(c_h))*u*v*dx = (b.grad c_h)|_K M_K
        (advection_solidbody_FCT_PDECO_finaltime.py:187,215)."""
        cl = self._local(c_dof)
        gc = np.einsum("ta,tad->td", cl, self.grad)
        s = gc[:, 0] * b[0] + gc[:, 1] * b[1]
        return self._mat(s[:, None, None] * self.MK)

    def drift2(self, c_dof, b=(1.0, 1.0)):
        """dot(drift, grad(v))*c_h*u*dx: A[i,j] = (b.grad phi_i)(M_K c_K)_j
        (advection_solidbody_FCT_PDECO_finaltime.py:188,216)."""
        cl = self._local(c_dof)
        bg = self.grad[:, :, 0] * b[0] + self.grad[:, :, 1] * b[1]  # (nt,3)
        Mc = np.einsum("tmj,tm->tj", self.MK, cl)
        return self._mat(bg[:, :, None] * Mc[:, None, :])

    def drift_gradient(self, p_dof, u_dof, b=(1.0, 1.0)):
        """assemble(p_h*dot(drift, grad(u_h))*v*dx) = (b.grad u_h)|_K (M_K p_K)_i
        (advection_solidbody_FCT_PDECO_finaltime.py:235-236)."""
        ul = self._local(u_dof)
        pl = self._local(p_dof)
        gu = np.einsum("ta,tad->td", ul, self.grad)
        s = gu[:, 0] * b[0] + gu[:, 1] * b[1]
        be = s[:, None] * np.einsum("tim,tm->ti", self.MK, pl)
        return self._vec(be)

    # -- chemotaxis exp-forms ---------------------------------------------
    def chtxs_forward_Aa(self, u_dof, v_dof, eta):
        """exp(-eta*u_n)*dot(grad(v_np1), grad(v))*u*dx (helpers.py:1350-1351):
        A[i,j] = int e^{-eta u_h} (grad v_h . grad phi_i) phi_j, 6-point rule."""
        pts, w = QUAD6_PTS, QUAD6_W
        uq = self._at(self._local(u_dof), pts)
        vl = self._local(v_dof)
        gv = np.einsum("ta,tad->td", vl, self.grad)
        gvi = np.einsum("td,tid->ti", gv, self.grad)  # grad v . grad phi_i
        e = np.exp(-eta * uq)
        Ke = np.einsum("t,q,tq,ti,qj->tij", self.area, w, e, gvi, pts)
        return self._mat(Ke)

    def chtxs_adjoint_Aa(self, u_dof, v_dof, eta):
        """(1-eta*u_n)*exp(-eta*u_n)*dot(grad(p), grad(v_n))*w*dx
        (helpers.py:1499-1500): A[i,j] = int (1-eta u)e^{-eta u}(grad phi_j . grad v_h) phi_i.
        UFL degree estimate 5 -> 7-point rule [inferred, parity unpinned]."""
        pts, w = QUAD7_PTS, QUAD7_W
        uq = self._at(self._local(u_dof), pts)
        vl = self._local(v_dof)
        gv = np.einsum("ta,tad->td", vl, self.grad)
        gvj = np.einsum("td,tjd->tj", gv, self.grad)
        f = (1.0 - eta * uq) * np.exp(-eta * uq)
        Ke = np.einsum("t,q,tq,qi,tj->tij", self.area, w, f, pts, gvj)
        return self._mat(Ke)

    def chtxs_adjoint_rhs_q(self, u_dof, p_dof, chi, eta):
        """chi*u_n*exp(-eta*u_n)*dot(grad(p_n), grad(w))*dx (helpers.py:1531-1532).
        UFL degree estimate 4 -> 6-point rule [inferred, parity unpinned]."""
        pts, w = QUAD6_PTS, QUAD6_W
        uq = self._at(self._local(u_dof), pts)
        pl = self._local(p_dof)
        gp = np.einsum("ta,tad->td", pl, self.grad)
        gpi = np.einsum("td,tid->ti", gp, self.grad)
        f = chi * uq * np.exp(-eta * uq)
        be = np.einsum("t,q,tq,ti->ti", self.area, w, f, gpi)
        return self._vec(be)


def row_lump_diag(M) -> np.ndarray:
    """Diagonal of row_lump(M) (helpers.py:309-328)."""
    return np.asarray(M.sum(axis=1)).ravel()
