"""Drop-in mirror of the reference's ``helpers.py`` operator API, backed by libfemfct.

Same names, argument order/meaning, return types and error behaviour as the
reference; every function runs on the GPU through the C ABI (no CPU fallback).
NumPy arrays in, NumPy arrays out; matrices are ``scipy.sparse`` of any format.

    FCT_alg_ref              helpers.py:1715-1872
    FCT_alg                  old_helpers.py:115-203 (old sign convention)
    ChebSI                   helpers.py:143-185
    artificial_diffusion_mat helpers.py:206-242
    row_lump                 helpers.py:309-328
    sparse_nonzero           helpers.py:187-204
    L2_norm_sq_Q/_Omega      helpers.py:330-381
    cost_functional          helpers.py:383-441
    rel_err                  helpers.py:69-85
    reorder_vector_to_dof / reorder_vector_from_dof (+ *_time aliases)  helpers.py:13-67
    find_node_neighbours     helpers.py:271-307 (takes the SquareMeshP1 descriptor)
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix, lil_matrix

from . import _lib
from .device import Context
from .mesh import (SquareMeshP1, reorder_vector_to_dof, reorder_vector_from_dof,  # noqa: F401
                   reorder_vector_to_dof_time, reorder_vector_from_dof_time)

VERBOSE = True  # print the reference's stdout diagnostics (helpers.py:1798-1809)

_DEVICE_ID = 0


def set_device(device_id: int):
    """GPU used by the module-level drop-in functions (one process per GPU)."""
    global _DEVICE_ID
    if int(device_id) != _DEVICE_ID:
        _PatternCache.clear()
    _DEVICE_ID = int(device_id)


class _Pattern:
    """A Context with the sparsity pattern (and mass matrix) of one M registered."""

    def __init__(self, M):
        Mc = csr_matrix(M)
        Mc.sum_duplicates()
        Mc.sort_indices()
        self.n = Mc.shape[0]
        self.indptr = Mc.indptr.astype(np.int32)
        self.indices = Mc.indices.astype(np.int32)
        self.nnz = self.indices.size
        self.rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.indptr))
        self.keys = self.rows * self.n + self.indices
        self.ctx = Context(_DEVICE_ID)
        self.ctx.set_pattern_csr(self.indptr, self.indices)
        self.M_data = None
        self.ml = None

    def set_mass(self, M, ml):
        m = self.values(M)
        ml = np.ascontiguousarray(ml, dtype=np.float64)
        if self.M_data is None or not (np.array_equal(m, self.M_data) and np.array_equal(ml, self.ml)):
            self.ctx.set_mass(m, ml)
            self.M_data, self.ml = m.copy(), ml.copy()

    def values(self, A) -> np.ndarray:
        """CSR values of A laid out on this pattern; nonzeros outside it are an error."""
        Ac = csr_matrix(A)
        if Ac.shape != (self.n, self.n):
            raise ValueError(f"matrix shape {Ac.shape} does not match the pattern ({self.n},{self.n})")
        Ac.sum_duplicates()
        Ac.sort_indices()
        if Ac.indices.size == self.nnz and np.array_equal(Ac.indptr, self.indptr) \
                and np.array_equal(Ac.indices, self.indices):
            return np.ascontiguousarray(Ac.data, dtype=np.float64)
        arows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(Ac.indptr))
        ka = arows * self.n + Ac.indices
        pos = np.searchsorted(self.keys, ka)
        ok = pos < self.nnz
        ok[ok] &= self.keys[pos[ok]] == ka[ok]
        if not np.all(ok | (Ac.data == 0)):
            raise ValueError("matrix has nonzeros outside the sparsity pattern of M")
        out = np.zeros(self.nnz)
        out[pos[ok]] = Ac.data[ok]
        return out

    def csr(self, vals):
        return csr_matrix((vals, self.indices, self.indptr), shape=(self.n, self.n))


class _PatternCacheT(dict):
    def get_for(self, M) -> _Pattern:
        Mc = M if isinstance(M, csr_matrix) else csr_matrix(M)
        if not Mc.has_sorted_indices:
            Mc = Mc.sorted_indices()
        key = (Mc.shape[0], Mc.nnz, hash(Mc.indptr.tobytes()), hash(Mc.indices.tobytes()))
        p = self.get(key)
        if p is None:
            if len(self) >= 4:
                self.clear()
            p = _Pattern(Mc)
            self[key] = p
        return p


_PatternCache = _PatternCacheT()


def _diag(mat) -> np.ndarray:
    return np.asarray(csr_matrix(mat).diagonal(), dtype=np.float64)


def FCT_alg_ref(A, rhs, u_n, dt, nodes, M, M_lumped, dof_neighbors, non_flux_mat=None,
                vertex_to_dof=None, info=None):
    """One linearised FEM-FCT backward-Euler step of
    ``[M + dt*(A + non_flux_mat)] u^{n+1} = M u^n + dt*rhs`` (helpers.py:1715-1872).

    Returns a new array; inputs are not mutated.  ``dof_neighbors`` is accepted
    for signature parity: the neighbour relation is the sparsity pattern of
    ``M`` (identical on a P1 mesh).  ``info`` (optional dict) receives the
    low-order solver diagnostics.
    """
    pat = _PatternCache.get_for(M)
    if nodes != pat.n:
        raise ValueError(f"nodes={nodes} does not match M ({pat.n})")
    pat.set_mass(M, _diag(M_lumped))
    a = pat.values(A)
    nn = None if non_flux_mat is None else pat.values(non_flux_mat)
    if np.isscalar(rhs):
        rhs = np.full(pat.n, float(rhs))
    u, inf = pat.ctx.fct_step_host(a, rhs, u_n, dt, N_csr_vals=nn)
    if info is not None:
        info.update(inf)
    if VERBOSE and (inf["flags"] & _lib.FLAG_MMATRIX_ROWSUM):
        # the reference's stdout diagnostic (helpers.py:1798-1809)
        print("3:", False)
        rs = np.asarray(pat.csr(a).sum(axis=1)).ravel()
        ml = pat.ml
        up = [-ml[i] / rs[i] for i in range(pat.n) if rs[i] < 0]
        lo = [-ml[i] / rs[i] for i in range(pat.n) if rs[i] > 0]
        print("Upper bound on dt:", min(up))
        print("Lower bound on dt:", max(max(lo), 0))
    return u


def FCT_alg(A, rhs, u_n, dt, nodes, M, M_lumped, dof_neighbors, source_mat=None):
    """Old sign convention used by the monolithic scripts (old_helpers.py:115-203):
    ``M u' = A u - source_mat u + rhs``  ==  ``FCT_alg_ref(-A, ..., non_flux_mat=source_mat)``."""
    return FCT_alg_ref(-csr_matrix(A), rhs, u_n, dt, nodes, M, M_lumped, dof_neighbors,
                       non_flux_mat=source_mat)


def ChebSI(vec, M, Md, cheb_iter=20, lmin=0.5, lmax=2):
    """Chebyshev semi-iteration for ``M x = vec`` (helpers.py:143-185).  ``Md`` is the preconditioner diagonal:
    ``M.diagonal()`` in every call the reference makes (helpers.py:1815; fused multi-sweep kernels), any other
    positive vector goes through the one-sweep kernels with ``Md`` as a device vector."""
    pat = _PatternCache.get_for(M)
    md = np.ascontiguousarray(Md, dtype=np.float64).ravel()
    if md.shape[0] != pat.n:
        raise ValueError(f"ChebSI: Md has {md.shape[0]} entries, M has {pat.n} rows")
    pat.set_mass(M, np.asarray(csr_matrix(M).sum(axis=1)).ravel())
    ctx = pat.ctx
    b = ctx.array(np.asarray(vec, dtype=np.float64).ravel())
    y = ctx.empty(pat.n)
    mdd = None
    try:
        if np.array_equal(md, _diag(M)):
            ctx.chebsi(b, y, cheb_iter, lmin, lmax)
        else:
            mdd = ctx.array(md)
            ctx.chebsi_md(b, y, mdd, cheb_iter, lmin, lmax)
        return y.download()
    finally:
        b.free()
        y.free()
        if mdd is not None:
            mdd.free()


def artificial_diffusion_mat(mat):
    """``d_ij = max(0, -k_ij, -k_ji)``, ``d_ii = -sum_j d_ij`` (helpers.py:206-242); LIL result."""
    K = csr_matrix(mat)
    S = (abs(K) + abs(K.T)).tocsr()     # symmetrised pattern
    S = (S + csr_matrix((np.ones(K.shape[0]), (np.arange(K.shape[0]), np.arange(K.shape[0]))), shape=K.shape)).tocsr()
    pat = _PatternCache.get_for(S)
    ctx = pat.ctx
    k = ctx.csr_to_ell(pat.values(K))
    d = ctx.empty(ctx.W * ctx.n)
    try:
        ctx.artificial_diffusion(k, d)
        vals = ctx.ell_to_csr(d, pat.nnz)
    finally:
        k.free()
        d.free()
    return lil_matrix(pat.csr(vals))


def row_lump(mat, nodes):
    """helpers.py:309-328 (host: one row sum at set-up time, not on the per-step path)."""
    out = lil_matrix((nodes, nodes))
    out.setdiag(np.asarray(csr_matrix(mat).sum(axis=1)).ravel())
    return out


def sparse_nonzero(H):
    """helpers.py:187-204: rows [row, col, value, value > 0]."""
    Hx = csr_matrix(H).tocoo()
    return np.transpose(np.array([Hx.row, Hx.col, Hx.data, Hx.data > 0]))


def rel_err(new, old):
    """helpers.py:69-85."""
    return np.linalg.norm(new - old) / np.linalg.norm(old)


def find_node_neighbours(mesh: SquareMeshP1, nodes=None, vertex_to_dof=None):
    """helpers.py:271-307 for the structured mesh descriptor."""
    return mesh.dof_neighbors()


# ---------------------------------------------------------------------------
# norms / cost functional (device reductions)
# ---------------------------------------------------------------------------
def _with_mass(M):
    pat = _PatternCache.get_for(M)
    pat.set_mass(M, np.asarray(csr_matrix(M).sum(axis=1)).ravel())
    return pat


def L2_norm_sq_Q(phi, num_steps, dt, M):
    """helpers.py:330-360."""
    pat = _with_mass(M)
    phi = np.asarray(phi, dtype=np.float64).ravel()
    if phi.size != (num_steps + 1) * pat.n:
        raise ValueError("array split does not result in an equal division")  # np.split's error (helpers.py:354)
    a = pat.ctx.array(phi)
    try:
        return float(pat.ctx.l2_norm_sq_Q(a, None, num_steps, dt)[0])
    finally:
        a.free()


def L2_norm_sq_Omega(phi, M):
    """helpers.py:362-381."""
    pat = _with_mass(M)
    a = pat.ctx.array(np.asarray(phi, dtype=np.float64).ravel())
    try:
        return float(pat.ctx.l2_norm_sq_Omega(a, None)[0])
    finally:
        a.free()


def cost_functional(var1, var1_target, projected_control, num_steps, dt, M, beta, optim,
                    var2=None, var2_target=None):
    """helpers.py:383-441."""
    valid_options = ["alltime", "finaltime"]
    if optim not in valid_options:
        raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of {valid_options}.")
    pat = _with_mass(M)
    ctx = pat.ctx
    bufs = []

    def up(x):
        d = ctx.array(np.asarray(x, dtype=np.float64).ravel())
        bufs.append(d)
        return d

    try:
        two = var2 is not None and var2_target is not None
        J = ctx.cost_functional(up(var1), up(var1_target), up(projected_control), num_steps, dt, beta, optim,
                                var2=up(var2) if two else None, var2_target=up(var2_target) if two else None)
        return float(J[0])
    finally:
        for d in bufs:
            d.free()
