"""Drop-in mirrors of the reference's trajectory solvers (same names, argument order and
in-place mutation behaviour), running as device-resident sweeps through the C ABI:

    solve_nonlinear_equation / solve_adjoint_nonlinear_equation   helpers.py:881-1038
    solve_schnak_system      / solve_adjoint_schnak_system        helpers.py:511-698
    solve_chtxs_system       / solve_adjoint_chtxs_system         helpers.py:1250-1581
    get_*_params / *_IC                                           helpers.py:443-509, 835-879, 1197-1248

``V`` is the ``SquareMeshP1`` descriptor (stand-in for the dolfin FunctionSpace); a ``control_fun``
is a Python float (the reference passes a constant dolfin Expression for target generation).
Reference quirks are reproduced: the forward solvers freeze the control at time
level 1 (helpers.py:577-578, 950-951, 1332-1333) and zero ``var[nodes:]`` in place first.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import _lib
from .device import Context
from .mesh import SquareMeshP1, reorder_vector_to_dof


# ----------------------------------------------------------------------------- parameters
def schnak_wind(x, y, t=0):
    """helpers.py:506-508 (stationary: the ``t`` parameter does not enter)."""
    return 1 * (y - 0.5) * x * (1 - x), -1 * (x - 0.5) * y * (1 - y)


def get_schnak_sys_params():
    """helpers.py:485-509: Du, Dv, c_a, c_b, gamma, omega1, omega2, wind."""
    return 1 / 100, 8.6676, 0.1, 0.9, 230.82, 100, 0.6, schnak_wind


def nonlinear_wind(x, y, speed=1):
    """helpers.py:876-878."""
    return speed * 2 * (y - 0.5) * x * (1 - x), -speed * 2 * (x - 0.5) * y * (1 - y)


def get_nonlinear_eqns_params():
    """helpers.py:867-879: eps, speed, wind."""
    return 1e-4, 1, nonlinear_wind


def get_chtxs_sys_params():
    """helpers.py:1197-1211: delta, Dm, Df, chi, gamma, eta."""
    return 100, 0.05, 0.05, 0.25, 100, 0.5


def _grid(a1, a2, deltax):
    X = np.arange(a1, a2 + deltax, deltax)
    return np.meshgrid(X, X)


def schnak_sys_IC(a1, a2, deltax, nodes, vertex_to_dof):
    """helpers.py:443-483."""
    X, Y = _grid(a1, a2, deltax)
    _, _, c_a, c_b, _, _, _, _ = get_schnak_sys_params()
    con = 0.1
    pert = 0.01 * (sum(np.cos(2 * np.pi * X * i) for i in range(1, 9)))
    u_init = c_a + c_b + con * np.cos(2 * np.pi * (X + Y)) + pert
    v_init = c_b / pow(c_a + c_b, 2) + con * np.cos(2 * np.pi * (X + Y)) + pert
    return (reorder_vector_to_dof(u_init.reshape(nodes), 1, nodes, vertex_to_dof),
            reorder_vector_to_dof(v_init.reshape(nodes), 1, nodes, vertex_to_dof))


def nonlinear_equation_IC(a1, a2, deltax, nodes, vertex_to_dof):
    """helpers.py:835-865."""
    X, Y = _grid(a1, a2, deltax)
    init = 5 * Y * (Y - 1) * X * (X - 1) * np.sin(4 * X * np.pi)
    return reorder_vector_to_dof(init.reshape(nodes), 1, nodes, vertex_to_dof)


def chtxs_sys_IC(a1, a2, deltax, nodes, vertex_to_dof):
    """helpers.py:1213-1248 (np.random.seed(5); v0 = u0)."""
    sq = round(np.sqrt(nodes))
    np.random.seed(5)
    u_init = 1.5 + 0.1 * (0.5 - np.random.rand(sq, sq))
    u0 = reorder_vector_to_dof(u_init.reshape(nodes), 1, nodes, vertex_to_dof)
    return u0, u0


# ----------------------------------------------------------------------------- device problem
class PDESystems:
    """One GPU context on a structured mesh with the convection matrices of the reference's
    winds assembled; exposes the device-resident sweeps."""

    def __init__(self, mesh: SquareMeshP1, device_id=0, order=_lib.ORDER_FENICS):
        self.mesh = mesh
        self.ctx = Context(device_id)
        self.ctx.set_mesh_square(mesh.a1, mesh.a2, mesh.n_cells, order)
        self.n = self.ctx.n
        self._conv = {}

    MAX_UNNAMED_WINDS = 8

    def convection(self, wind, name=None):
        """assemble_sparse(dot(wind, grad(v))*u*dx) on the device (cached per wind).

        Named winds (the reference's parameter sets) stay for the life of the system.  A wind passed as a bare function
        is keyed on the function object; a caller that builds a new lambda per call would otherwise add two device
        matrices per call to a long-lived system, so only the MAX_UNNAMED_WINDS most recent ones are kept."""
        key = name if name is not None else wind       # the function object itself: stays alive, never aliases
        if key not in self._conv:
            if name is None:
                unnamed = [k for k in self._conv if callable(k)]
                for old in unnamed[:max(0, len(unnamed) - self.MAX_UNNAMED_WINDS + 1)]:
                    self.ctx.synchronize()
                    for arr in self._conv.pop(old):
                        arr.free()
            xq, yq = self.ctx.quad_points(self.mesh.n_cells)
            wx, wy = wind(xq, yq)
            A = self.ctx.assemble_convection(np.stack([wx, wy], axis=1).reshape(-1))
            self._conv[key] = (A, self.ctx.ell_transpose(A))
        return self._conv[key]

    def close(self):
        self.ctx.close()


_cache = {}


def _system(V: SquareMeshP1) -> PDESystems:
    if not isinstance(V, SquareMeshP1):
        raise TypeError("V must be a SquareMeshP1 mesh descriptor (stand-in for the dolfin FunctionSpace)")
    s = _cache.get(V.key())
    if s is None:
        if len(_cache) >= 2:
            for old in _cache.values():
                old.close()
            _cache.clear()
        # the device works in vertex (lexicographic) order, where the index-free tile kernels and the
        # Chebyshev species solve apply; _Bufs permutes FEniCS-ordered host vectors at the boundary
        s = PDESystems(V, order=_lib.ORDER_VERTEX)
        s.v2d = np.asarray(V.vertex_to_dof, dtype=np.int64)
        _cache[V.key()] = s
    return s


def _frozen_control(control, control_fun, nodes):
    if control_fun is not None:
        return np.full(nodes, float(control_fun))
    return np.array(control[nodes:2 * nodes], dtype=np.float64)   # level 1 for every step


class _Bufs:
    """Host (FEniCS DoF order, level-major) <-> device (vertex order) staging of one call."""

    def __init__(self, S):
        self.ctx, self.items, self.v2d, self.n = S.ctx, [], S.v2d, S.n

    def up(self, x):
        x = np.asarray(x, dtype=np.float64).ravel()
        d = self.ctx.array(np.ascontiguousarray(x.reshape(-1, self.n)[:, self.v2d]).ravel())
        self.items.append(d)
        return d

    def down(self, d, out):
        """device vector(s) -> ``out`` (in place), back in FEniCS DoF order"""
        tmp = np.empty((d.count // self.n, self.n))
        tmp[:, self.v2d] = d.download().reshape(-1, self.n)
        out[...] = tmp.reshape(out.shape)
        return out

    def zeros(self, count):
        d = self.ctx.zeros(count)
        self.items.append(d)
        return d

    def free(self):
        for d in self.items:
            d.free()


# ----------------------------------------------------------------------------- nonlinear
def solve_nonlinear_equation(control, var1, var2, V, nodes, num_steps, dt, dof_neighbors,
                             control_fun=None, show_plots=False, vertex_to_dof=None):
    """helpers.py:881-966: mutates ``var1[nodes:]`` in place, returns ``(var1, None)``."""
    if var2 is not None:
        warnings.warn("Warning: 'var2' is not None. Ensure this is intentional.")
    S = _system(V)
    eps, _, wind = get_nonlinear_eqns_params()
    Aw, _ = S.convection(wind, "nonlinear")
    var1[nodes:] = np.zeros(num_steps * nodes)
    B = _Bufs(S)
    try:
        u = B.up(var1)
        S.ctx.nonlinear_forward(Aw, B.up(_frozen_control(control, control_fun, nodes)), u, num_steps, dt, eps)
        B.down(u, var1)
    finally:
        B.free()
    return var1, None


def solve_adjoint_nonlinear_equation(uk, uhat_T, pk, T, V, nodes, num_steps, dt, dof_neighbors):
    """helpers.py:968-1038: fills ``pk`` (terminal condition included) and returns it."""
    S = _system(V)
    eps, _, wind = get_nonlinear_eqns_params()
    Aw, _ = S.convection(wind, "nonlinear")
    B = _Bufs(S)
    try:
        p = B.up(pk)
        S.ctx.nonlinear_adjoint(Aw, B.up(uk), B.up(uhat_T), p, num_steps, dt, eps)
        B.down(p, pk)
    finally:
        B.free()
    return pk


# ----------------------------------------------------------------------------- Schnakenberg
def _schnak_par():
    Du, Dv, _, c_b, gamma, omega1, omega2, wind = get_schnak_sys_params()
    return [Du, Dv, c_b, gamma, omega1, omega2], wind


def _wind_factors(wind_scale, num_steps, dt, t0=0.0, T=None):
    """s(t_k), k = 0..num_steps, from a callable ``s(t)`` or an array; None stays None (stationary wind).

    The time levels are ACCUMULATED the way the reference's loops do it, so that ``s`` sees the same floating-point
    arguments: forward ``t = t0; t += dt`` per step (helpers.py:565-566); adjoint (``T`` given) ``t = T; t -= dt`` per
    step (helpers.py:664, 679) -- ``t0 + k*dt`` differs from both in the last bits, amplified by 2 pi in sin(2 pi t)."""
    if wind_scale is None:
        return None
    if callable(wind_scale):
        out = np.empty(num_steps + 1)
        if T is None:
            t = t0
            out[0] = float(wind_scale(t))
            for k in range(1, num_steps + 1):
                t += dt
                out[k] = float(wind_scale(t))
        else:
            t = T
            out[num_steps] = float(wind_scale(t))
            for k in range(num_steps - 1, -1, -1):
                t -= dt
                out[k] = float(wind_scale(t))
        return out
    ws = np.asarray(wind_scale, dtype=np.float64).ravel()
    if ws.size != num_steps + 1:
        raise ValueError(f"wind_scale: {ws.size} values, expected num_steps + 1 = {num_steps + 1}")
    return ws


def solve_schnak_system(control, var1, var2, V, nodes, num_steps, dt, dof_neighbors,
                        control_fun=None, rescaling=1, wind=None, wind_scale=None):
    """helpers.py:511-597: mutates and returns ``(var1, var2)``.

    Extension for the script BASELINE config 3 names (Schnak_FCT_PDECO_alltime.py:55,174-175: the wind
    ``(-(y-.5), (x-.5)) * sin(2 pi t)``, re-assembled per step): ``wind=w0`` (a function ``(x, y) -> (wx, wy)``,
    default the stationary wind of helpers.py:506-508) and ``wind_scale=s`` (callable ``s(t)`` or the array
    ``s(t_0..t_Nt)``) give the separable wind ``s(t) w0(x)``; the step to level n+1 uses ``s(t_{n+1})``
    (helpers.py:565-566: ``t += dt; wind.t = t``)."""
    S = _system(V)
    par, wind0 = _schnak_par()
    Aw, _ = S.convection(wind or wind0, None if wind is not None else "schnak")
    var1[nodes:] = np.zeros(num_steps * nodes)
    var2[nodes:] = np.zeros(num_steps * nodes)
    B = _Bufs(S)
    try:
        u, v = B.up(var1), B.up(var2)
        S.ctx.schnak_forward(Aw, B.up(_frozen_control(control, control_fun, nodes)), u, v, num_steps, dt, par, rescaling,
                             wind_scale=_wind_factors(wind_scale, num_steps, dt))
        B.down(u, var1)
        B.down(v, var2)
    finally:
        B.free()
    return var1, var2


def solve_adjoint_schnak_system(uk, vk, uhat_T, vhat_T, pk, qk, T, V, nodes, num_steps, dt, dof_neighbors,
                                optim="finaltime", wind=None, wind_scale=None):
    """helpers.py:599-698 (``optim="finaltime"``, the reference's signature and behaviour).
    ``optim="alltime"`` (extension, structure of Schnak_FCT_PDECO_alltime.py:204-284): the targets are
    trajectories, p(T) = q(T) = 0 and both equations carry the assembled misfit.
    ``wind`` / ``wind_scale``: as in :func:`solve_schnak_system`; the step that produces level n uses ``s(t_n)``
    (helpers.py:664, 679: ``t -= dt; wind.t = t``)."""
    if optim not in ("alltime", "finaltime"):
        raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of ['alltime', 'finaltime'].")
    S = _system(V)
    par, wind0 = _schnak_par()
    _, AwT = S.convection(wind or wind0, None if wind is not None else "schnak")
    B = _Bufs(S)
    try:
        p, q = B.up(pk), B.up(qk)
        S.ctx.schnak_adjoint(AwT, B.up(uk), B.up(vk), B.up(uhat_T), B.up(vhat_T), p, q, num_steps, dt, par,
                             alltime=optim == "alltime", wind_scale=_wind_factors(wind_scale, num_steps, dt, T=T))
        B.down(p, pk)
        B.down(q, qk)
    finally:
        B.free()
    return pk, qk


# ----------------------------------------------------------------------------- chemotaxis
def _chtxs_par():
    delta, Dm, Df, chi, _, eta = get_chtxs_sys_params()
    return [delta, Dm, Df, chi, eta]


def solve_chtxs_system(control, var1, var2, V, nodes, num_steps, dt, dof_neighbors,
                       control_fun=None, show_plots=False, vertex_to_dof=None,
                       generation_mode=False, output_dir=None, rescaling=1 / 10):
    """helpers.py:1250-1385."""
    S = _system(V)
    par = _chtxs_par()
    B = _Bufs(S)
    try:
        if generation_mode:
            if len(var1) != nodes or len(var2) != nodes or len(control) != nodes:
                raise ValueError(f"Generation mode, the input vectors should be of length {nodes}")
            tl = (num_steps + 1) * nodes
            u0 = np.zeros(tl)
            v0 = np.zeros(tl)
            u0[:nodes], v0[:nodes] = var1, var2
            c_level = np.full(nodes, float(control_fun)) if control_fun is not None else np.asarray(control, dtype=float)
            u, v = B.up(u0), B.up(v0)
            S.ctx.chtxs_forward(B.up(c_level), u, v, num_steps, dt, par, rescaling)
            if output_dir is not None:
                uu, vv = B.down(u, np.empty(tl)), B.down(v, np.empty(tl))
                t = 0
                for i in range(1, num_steps + 1):
                    t += dt
                    if i % 100 == 0:   # helpers.py:1363-1367
                        uu[i * nodes:(i + 1) * nodes].tofile(output_dir / f"chtxs_m_t{round(t, 2)}.csv", sep=",")
                        vv[i * nodes:(i + 1) * nodes].tofile(output_dir / f"chtxs_f_t{round(t, 2)}.csv", sep=",")
            return var1, var2       # the reference returns its (unchanged) inputs in this mode
        var1[nodes:] = np.zeros(num_steps * nodes)
        var2[nodes:] = np.zeros(num_steps * nodes)
        u, v = B.up(var1), B.up(var2)
        S.ctx.chtxs_forward(B.up(_frozen_control(control, control_fun, nodes)), u, v, num_steps, dt, par, rescaling)
        B.down(u, var1)
        B.down(v, var2)
    finally:
        B.free()
    return var1, var2


def solve_adjoint_chtxs_system(uk, vk, uhat, vhat, pk, qk, control, T, V, nodes, num_steps, dt,
                               dof_neighbors, optim, show_plots=None, vertex_to_dof=None, out_folder=None,
                               mesh=None, deltax=None, rescaling=1 / 10):
    """helpers.py:1387-1581."""
    valid_options = ["alltime", "finaltime"]
    if optim not in valid_options:
        raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of {valid_options}.")
    S = _system(V)
    B = _Bufs(S)
    try:
        p, q = B.up(pk), B.up(qk)
        S.ctx.chtxs_adjoint(B.up(uk), B.up(vk), B.up(uhat), B.up(vhat), p, q, B.up(control), num_steps, dt,
                            _chtxs_par(), rescaling, optim == "alltime")
        B.down(p, pk)
        B.down(q, qk)
    finally:
        B.free()
    return pk, qk


# ----------------------------------------------------------------------------- assembly bridge
def assemble_mass(V: SquareMeshP1):
    """``assemble_sparse(u*v*dx)`` (helpers.py:87-104, 553): the device-assembled mass matrix as a
    scipy CSR matrix in FEniCS DoF order."""
    from scipy.sparse import csr_matrix
    S = _system(V)
    ctx = S.ctx
    cols = ctx.ell_cols()
    n, W = ctx.n, ctx.W
    rows = np.tile(np.arange(n), W)
    flat = cols.reshape(-1)
    mask = (flat != rows) | (np.arange(W * n) < n)
    r, c = rows[mask], flat[mask]
    order = np.lexsort((c, r))
    vals = ctx.ell_to_csr(ctx.mass_ell, int(mask.sum()))
    indptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))])
    Mv = csr_matrix((vals, c[order], indptr), shape=(n, n))
    return ell_matrix_to_dof_order(Mv, S.v2d)


def ell_matrix_to_dof_order(Mv, v2d):
    """device (vertex) ordering -> FEniCS DoF ordering of a sparse matrix: M_dof[v2d[i], v2d[j]] = M_v[i, j]"""
    d2v = np.empty_like(v2d)
    d2v[v2d] = np.arange(v2d.size)
    out = Mv.tocsr()[d2v][:, d2v].tocsr()
    out.sort_indices()
    return out


def device_matrix(V: SquareMeshP1, ell):
    """A device ELL matrix of the mesh's context (e.g. ``_system(V).convection(wind)[0]``) as scipy CSR in
    FEniCS DoF order -- what ``assemble_sparse`` would have returned."""
    from scipy.sparse import csr_matrix
    S = _system(V)
    ctx = S.ctx
    cols = ctx.ell_cols()
    n, W = ctx.n, ctx.W
    rows = np.tile(np.arange(n), W)
    flat = cols.reshape(-1)
    mask = (flat != rows) | (np.arange(W * n) < n)
    r, c = rows[mask], flat[mask]
    order = np.lexsort((c, r))
    vals = ctx.ell_to_csr(ell, int(mask.sum()))
    indptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))])
    return ell_matrix_to_dof_order(csr_matrix((vals, c[order], indptr), shape=(n, n)), S.v2d)


# ----------------------------------------------------------------------------- line search
def armijo_line_search_ref(var1, c, d, var1_target, num_steps, dt, c_lower, c_upper, beta, costfun_init,
                           nodes, optim, V, gam=1e-4, max_iter=10, s0=1, nonlinear_solver=None,
                           dof_neighbors=None, var2=None, var2_target=None, w1=None, w2=None):
    """Projected Armijo line search, helpers.py:1583-1713 (live branch: ``w1 is None``; with ``w1``
    given the reference leaves ``M = None`` and fails inside cost_functional, :1654-1663).
    The host loop stays in Python as in the reference; every trial's state solve and both
    norm evaluations run on the GPU.  Returns ``(var1[, var2], c_inc, k+1)``; the caller detects
    failure by ``k+1 == max_iter`` (nonlinear_FCT_PDECO_refactored.py:161)."""
    from .fct_helpers import cost_functional, L2_norm_sq_Q
    valid_options = ["alltime", "finaltime"]
    if optim not in valid_options:
        raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of {valid_options}.")
    if w1 is not None or w2 is not None:
        raise ValueError("armijo_line_search_ref: the linear-increment branch (w1/w2) is dead in the "
                         "reference (M is None there); pass nonlinear_solver instead")
    M = assemble_mass(V)
    s = s0
    armijo = float("inf")
    control_dif_L2 = 1
    k = 0
    c_inc = c
    for k in range(max_iter):
        c_inc = np.clip(c + s * d, c_lower, c_upper)
        var1, var2 = nonlinear_solver(c_inc, var1, var2, V, nodes, num_steps, dt, dof_neighbors)
        cost2 = cost_functional(var1, var1_target, c_inc, num_steps, dt, M, beta, optim=optim,
                                var2=var2, var2_target=var2_target)
        armijo = cost2 - costfun_init
        control_dif_L2 = L2_norm_sq_Q(c_inc - c, num_steps, dt, M)
        if armijo <= -gam / s * control_dif_L2:
            break
        s /= 2
    return (var1, var2, c_inc, k + 1) if var2 is not None else (var1, c_inc, k + 1)
