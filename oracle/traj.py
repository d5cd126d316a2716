"""CPU restatement of the reference's forward / adjoint trajectory solvers.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The dolfin ``FunctionSpace V`` argument of the reference is replaced by an
``oracle.assembly.P1Assembler`` (``asm``); every other argument keeps the
reference's position, meaning and in-place mutation behaviour.

  solve_nonlinear_equation          helpers.py:881-966
  solve_adjoint_nonlinear_equation  helpers.py:968-1038
  solve_schnak_system               helpers.py:511-597
  solve_adjoint_schnak_system       helpers.py:599-698
  solve_chtxs_system                helpers.py:1250-1385
  solve_adjoint_chtxs_system        helpers.py:1387-1581
  solidbody_forward / _adjoint / _descent_direction
        advection_solidbody_FCT_PDECO_finaltime.py:175-193, 204-221, 228-238
        advection_solidbody_FCT_PDECO_alltime.py:239-259 (all-time rhs)

Reference quirks reproduced on purpose:
  1. forward solvers freeze the control at time level 1
     (helpers.py:577-578, 950-951, 1332-1333);
  2. all-time chemotaxis adjoint adds raw nodal misfits (helpers.py:1506-1507,
     1533-1534);
  3. du/dt ignores non_flux_mat (inside fct_step).
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.linalg import spsolve

from .assembly import P1Assembler, row_lump_diag
from .fct import Pattern, fct_step, chebsi
from scipy.sparse import diags


# ---------------------------------------------------------------------------
# parameter providers (values are inputs: helpers.py:485-509, 867-879, 1197-1211)
# ---------------------------------------------------------------------------
def schnak_params():
    """helpers.py:498-508."""
    return dict(Du=1 / 100, Dv=8.6676, c_a=0.1, c_b=0.9, gamma=230.82, omega1=100, omega2=0.6)


def schnak_wind(x, y):
    """helpers.py:506-507 (stationary; the ``t`` parameter is unused)."""
    return 1 * (y - 0.5) * x * (1 - x), -1 * (x - 0.5) * y * (1 - y)


def nonlinear_params():
    """helpers.py:873-874."""
    return dict(eps=1e-4, speed=1)


def nonlinear_wind(x, y):
    """helpers.py:876-877."""
    return 1 * 2 * (y - 0.5) * x * (1 - x), -1 * 2 * (x - 0.5) * y * (1 - y)


def chtxs_params():
    """helpers.py:1205-1210."""
    return dict(delta=100, Dm=0.05, Df=0.05, chi=0.25, gamma=100, eta=0.5)


def rotation_wind(om):
    """advection_solidbody_FCT_PDECO_finaltime.py:91-93: 1/om * (-y, x)."""
    return lambda x, y: (-(1 / om) * y, (1 / om) * x)


class _Common:
    def __init__(self, asm: P1Assembler):
        self.asm = asm
        self.M = asm.mass()
        self.ml = row_lump_diag(self.M)
        self.ML = diags(self.ml).tocsr()
        self.Ad = asm.stiffness()
        self.pat = Pattern(self.M)

    def fct(self, A, rhs, u_n, dt, non_flux_mat=None, info=None):
        return fct_step(A, rhs, u_n, dt, self.asm.n, self.M, self.ML, None,
                        non_flux_mat=non_flux_mat, pattern=self.pat, info=info)


_cache: dict = {}


def _common(asm) -> _Common:
    c = _cache.get(id(asm))
    if c is None or c.asm is not asm:
        c = _Common(asm)
        _cache.clear()
        _cache[id(asm)] = c
    return c


# ---------------------------------------------------------------------------
# nonlinear advection-reaction equation
# ---------------------------------------------------------------------------
def solve_nonlinear_equation(control, var1, var2, asm, nodes, num_steps, dt, dof_neighbors=None,
                             control_const=None):
    """helpers.py:881-966.  ``control_const`` plays ``control_fun`` (a constant
    Expression used for target generation)."""
    cm = _common(asm)
    P = nonlinear_params()
    A = asm.convection(nonlinear_wind)
    Mat_var1 = A - P["eps"] * cm.Ad
    var1[nodes:] = np.zeros(num_steps * nodes)
    frozen = None
    for i in range(1, num_steps + 1):
        start, end = i * nodes, (i + 1) * nodes
        var1_n = var1[start - nodes:start]
        if frozen is None:  # quirk 1
            frozen = (np.full(nodes, float(control_const)) if control_const is not None
                      else control[start:end].copy())
        M_u2 = asm.weighted_mass(lambda at: at(var1_n) ** 2)
        Mat_rhs = -cm.M + 1 / 3 * M_u2
        rhs = asm.load(lambda at: at(frozen))
        var1[start:end] = cm.fct(-Mat_var1, rhs, var1_n, dt, non_flux_mat=Mat_rhs)
    return var1, None


def solve_adjoint_nonlinear_equation(uk, uhat_T, pk, T, asm, nodes, num_steps, dt, dof_neighbors=None):
    """helpers.py:968-1038."""
    cm = _common(asm)
    P = nonlinear_params()
    A = asm.convection(nonlinear_wind)
    Mat_p = -A - P["eps"] * cm.Ad
    pk[num_steps * nodes:] = uhat_T - uk[num_steps * nodes:]
    for i in reversed(range(0, num_steps)):
        start, end = i * nodes, (i + 1) * nodes
        pk_np1 = pk[end:end + nodes]
        uk_n = uk[start:end]
        M_u2 = asm.weighted_mass(lambda at: at(uk_n) ** 2)
        Mat_rhs = M_u2 - cm.M
        pk[start:end] = cm.fct(-Mat_p, np.zeros(nodes), pk_np1, dt, non_flux_mat=Mat_rhs)
    return pk


# ---------------------------------------------------------------------------
# advective Schnakenberg
# ---------------------------------------------------------------------------
def solve_schnak_system(control, var1, var2, asm, nodes, num_steps, dt, dof_neighbors=None,
                        control_const=None, rescaling=1, wind=None, wind_scale=None):
    """helpers.py:511-597.  wind / wind_scale: the separable time-dependent wind s(t) w0(x) of
    Schnak_FCT_PDECO_alltime.py:55,174-175 -- ``wind.t = t`` with t = t_{n+1} before the matrix is assembled
    (helpers.py:565-566); for the stationary HEAD wind the assignment has no effect."""
    cm = _common(asm)
    P = schnak_params()
    Du, Dv, c_b, gamma, om1, om2 = P["Du"], P["Dv"], P["c_b"], P["gamma"], P["omega1"], P["omega2"]
    var1[nodes:] = np.zeros(num_steps * nodes)
    var2[nodes:] = np.zeros(num_steps * nodes)
    A0 = asm.convection(wind or schnak_wind)  # wind.t has no effect on the HEAD wind (helpers.py:506-508)
    rhs_var2 = asm.load(lambda at: np.full_like(at(np.zeros(nodes)), gamma * c_b))
    frozen = None
    t = 0.0
    for i in range(1, num_steps + 1):
        start, end = i * nodes, (i + 1) * nodes
        t += dt
        A = A0 if wind_scale is None else float(wind_scale(t)) * A0      # assemble(dot(s(t) w0, grad(v))*u*dx)
        u_n = var1[start - nodes:start]
        v_n = var2[start - nodes:start]
        if frozen is None:  # quirk 1
            frozen = (np.full(nodes, float(control_const)) if control_const is not None
                      else control[start:end].copy())
        Mat_var1 = Du * cm.Ad - om1 * A
        rhs_var1 = asm.load(lambda at: gamma / rescaling * at(frozen) + gamma * (at(u_n) ** 2 * at(v_n)))
        var1[start:end] = cm.fct(Mat_var1, rhs_var1, u_n, dt, non_flux_mat=gamma * cm.M)
        u_np1 = var1[start:end]
        M_u2 = asm.weighted_mass(lambda at: at(u_np1) ** 2)
        Mat_var2 = cm.M + dt * (Dv * cm.Ad - om2 * A + gamma * M_u2)
        var2[start:end] = spsolve(Mat_var2.tocsc(), cm.M @ v_n + dt * rhs_var2)
    return var1, var2


def solve_adjoint_schnak_system(uk, vk, uhat_T, vhat_T, pk, qk, T, asm, nodes, num_steps, dt,
                                dof_neighbors=None, optim="finaltime", wind=None, wind_scale=None):
    """helpers.py:599-698 (optim="finaltime").  optim="alltime" is an extension without a HEAD counterpart:
    zero terminal conditions and the misfit loads of the inline loop Schnak_FCT_PDECO_alltime.py:268
    (rhs_q += assemble((vhat_n - v_n)*w*dx)) and :278 (rhs_p += assemble((uhat_n - u_n)*w*dx)), alpha = 1."""
    cm = _common(asm)
    P = schnak_params()
    Du, Dv, gamma, om1, om2 = P["Du"], P["Dv"], P["gamma"], P["omega1"], P["omega2"]
    alltime = optim == "alltime"
    if alltime:
        pk[num_steps * nodes:] = 0.0
        qk[num_steps * nodes:] = 0.0
    else:
        pk[num_steps * nodes:] = uhat_T - uk[num_steps * nodes:]
        qk[num_steps * nodes:] = vhat_T - vk[num_steps * nodes:]
    # dot(wind, grad(u))*w*dx is the transpose of dot(wind, grad(w))*u*dx (helpers.py:681)
    A0 = asm.convection(wind or schnak_wind).T.tocsr()
    t = T
    for i in reversed(range(0, num_steps)):
        start, end = i * nodes, (i + 1) * nodes
        t -= dt                                                            # helpers.py:664, 679: wind.t = t_n
        A = A0 if wind_scale is None else float(wind_scale(t)) * A0
        q_np1 = qk[end:end + nodes]
        p_np1 = pk[end:end + nodes]
        u_n = uk[start:end]
        v_n = vk[start:end]
        M_u2 = asm.weighted_mass(lambda at: at(u_n) ** 2)
        rhs_q = asm.load(lambda at: gamma * at(p_np1) * at(u_n) ** 2)
        if alltime:
            rhs_q = rhs_q + cm.M @ (vhat_T[start:end] - v_n)
        Mat_q = cm.M + dt * (Dv * cm.Ad - om2 * A + gamma * M_u2)
        qk[start:end] = spsolve(Mat_q.tocsc(), cm.M @ q_np1 + dt * rhs_q)
        q_n = qk[start:end]
        Mat_p = Du * cm.Ad - om1 * A
        M_uv = asm.weighted_mass(lambda at: at(u_n) * at(v_n))
        rhs_p = asm.load(lambda at: -2 * gamma * at(u_n) * at(v_n) * at(q_n))
        if alltime:
            rhs_p = rhs_p + cm.M @ (uhat_T[start:end] - u_n)
        Mat_rhs = gamma * cm.M - 2 * gamma * M_uv
        pk[start:end] = cm.fct(Mat_p, rhs_p, p_np1, dt, non_flux_mat=Mat_rhs)
    return pk, qk


# ---------------------------------------------------------------------------
# chemotaxis
# ---------------------------------------------------------------------------
def solve_chtxs_system(control, var1, var2, asm, nodes, num_steps, dt, dof_neighbors=None,
                       control_const=None, rescaling=1 / 10):
    """helpers.py:1250-1385 (non-generation mode)."""
    cm = _common(asm)
    P = chtxs_params()
    delta, Dm, Df, chi, eta = P["delta"], P["Dm"], P["Df"], P["chi"], P["eta"]
    Mat_var2 = (cm.M + dt * (Df * cm.Ad + delta * cm.M)).tocsc()
    var1[nodes:] = np.zeros(num_steps * nodes)
    var2[nodes:] = np.zeros(num_steps * nodes)
    frozen = None
    for i in range(1, num_steps + 1):
        start, end = i * nodes, (i + 1) * nodes
        u_n = var1[start - nodes:start]
        v_n = var2[start - nodes:start]
        if frozen is None:  # quirk 1
            frozen = (np.full(nodes, float(control_const)) if control_const is not None
                      else control[start:end].copy())
        rhs2 = asm.load(lambda at: at(v_n) + dt * at(frozen) * at(u_n) / rescaling)
        v_np1 = spsolve(Mat_var2, rhs2)
        var2[start:end] = v_np1
        Aa = asm.chtxs_forward_Aa(u_n, v_np1, eta)
        A_var1 = Dm * cm.Ad - chi * Aa
        var1[start:end] = cm.fct(A_var1, np.zeros(nodes), u_n, dt)
    return var1, var2


def solve_adjoint_chtxs_system(uk, vk, uhat, vhat, pk, qk, control, T, asm, nodes, num_steps, dt,
                               dof_neighbors=None, optim="alltime", rescaling=1 / 10):
    """helpers.py:1387-1581.  exp-forms use inferred quadrature degrees
    (parity unpinned, see oracle/assembly.py)."""
    if optim not in ("alltime", "finaltime"):
        raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of ['alltime', 'finaltime'].")
    cm = _common(asm)
    P = chtxs_params()
    delta, Dm, Df, chi, eta = P["delta"], P["Dm"], P["Df"], P["chi"], P["eta"]
    if optim == "finaltime":
        pk[num_steps * nodes:] = uhat - uk[num_steps * nodes:]
        qk[num_steps * nodes:] = vhat - vk[num_steps * nodes:]
    Mat_q = (cm.M + dt * (Df * cm.Ad + delta * cm.M)).tocsc()
    for i in reversed(range(0, num_steps)):
        start, end = i * nodes, (i + 1) * nodes
        q_np1 = qk[end:end + nodes]
        p_np1 = pk[end:end + nodes]
        u_n = uk[start:end]
        v_n = vk[start:end]
        c_n = control[start:end]  # refreshed every step here (helpers.py:1496)
        Aa = asm.chtxs_adjoint_Aa(u_n, v_n, eta)
        Mat_p = Dm * cm.Ad - chi * Aa
        rhs_p = asm.load(lambda at: at(c_n) * at(q_np1) / rescaling)
        if optim == "alltime":
            rhs_p = rhs_p + (uhat[start:end] - uk[start:end])  # quirk 2
        pk[start:end] = cm.fct(Mat_p, rhs_p, p_np1, dt)
        p_n = pk[start:end]
        rhs_q = asm.chtxs_adjoint_rhs_q(u_n, p_n, chi, eta)
        if optim == "alltime":
            rhs_q = rhs_q + (vhat[start:end] - vk[start:end])  # quirk 2
        qk[start:end] = spsolve(Mat_q, cm.M @ q_np1 + dt * rhs_q)
    return pk, qk


# ---------------------------------------------------------------------------
# solid-body rotation + drift control (inline loops of the advection scripts)
# ---------------------------------------------------------------------------
class SolidBody:
    """Operators of advection_solidbody_FCT_PDECO_{finaltime,alltime}.py."""

    def __init__(self, asm, om=np.pi / 40, eps=0.0, drift=(1.0, 1.0), rot_scale=1.0):
        self.asm = asm
        self.cm = _common(asm)
        self.eps = eps
        self.drift = drift
        # finaltime.py:122 ; alltime.py:146 multiplies Arot by 0 (rot_scale=0)
        self.Arot = rot_scale * asm.convection(rotation_wind(om))

    def A_u(self, c_level):
        """finaltime.py:187-191: A_u = -eps*Ad + Arot + Adrift1 + Adrift2."""
        return (-self.eps * self.cm.Ad + self.Arot + self.asm.drift1(c_level, self.drift)
                + self.asm.drift2(c_level, self.drift))


def solidbody_forward(sb: SolidBody, ck, uk, nodes, num_steps, dt):
    """finaltime.py:175-193 (old-sign FCT_alg(A_u,...) == FCT_alg_ref(-A_u,...))."""
    uk[nodes:] = np.zeros(num_steps * nodes)
    for i in range(1, num_steps + 1):
        start, end = i * nodes, (i + 1) * nodes
        u_n = uk[start - nodes:start]
        A_u = sb.A_u(ck[start:end])          # control at level n+1
        uk[start:end] = sb.cm.fct(-A_u, np.zeros(nodes), u_n, dt)
    return uk


def solidbody_adjoint(sb: SolidBody, ck, uk, uhat, pk, nodes, num_steps, dt, optim="finaltime"):
    """finaltime.py:200-221 / alltime.py:232-259.
    A_p = -eps*Ad - Arot - Adrift1 - Adrift2 with the control at level n."""
    pk[:] = 0.0
    if optim == "finaltime":
        pk[num_steps * nodes:] = uhat - uk[num_steps * nodes:]
    for i in reversed(range(0, num_steps)):
        start, end = i * nodes, (i + 1) * nodes
        p_np1 = pk[end:end + nodes]
        c_n = ck[start:end]
        A_p = (-sb.eps * sb.cm.Ad - sb.Arot - sb.asm.drift1(c_n, sb.drift) - sb.asm.drift2(c_n, sb.drift))
        if optim == "alltime":
            u_n = uk[start:end]
            uh_n = uhat[start:end]
            rhs = sb.asm.load(lambda at: at(uh_n) - at(u_n))  # alltime.py:257
        else:
            rhs = np.zeros(nodes)
        pk[start:end] = sb.cm.fct(-A_p, rhs, p_np1, dt)
    return pk


def solidbody_descent_direction(sb: SolidBody, ck, uk, pk, beta, nodes, num_steps):
    """finaltime.py:228-238: dk = ChebSI(-(beta*M*c + int p (b.grad u) v)) per level."""
    dk = np.zeros_like(ck)
    M = sb.cm.M
    Md = M.diagonal()
    for i in range(num_steps + 1):
        start, end = i * nodes, (i + 1) * nodes
        rhs = -(beta * (M @ ck[start:end]) + sb.asm.drift_gradient(pk[start:end], uk[start:end], sb.drift))
        dk[start:end] = chebsi(rhs, M, Md, 20, 0.5, 2)
    return dk


def solidbody_pgd_loop(sb: SolidBody, u0, uhat, c0, beta, c_lower, c_upper, iters, nodes, num_steps, dt,
                       gam=1e-4, s0=1.0, max_armijo=10, optim="finaltime"):
    """The projected-gradient loop of advection_solidbody_FCT_PDECO_finaltime_Garvie.py:164-330 (optim="finaltime":
    uhat = target at T) and ..._alltime_Garvie.py:164-340 (optim="alltime": uhat = target trajectory), with the inline
    Armijo search of :259-317: trial steps s0 / 2^k, the first with J(c_inc) - J_k <= -gam/s ||c_inc - c||^2_Q is taken,
    else the last.  Returns (u, p, c, history); history["armijo_margin"][it][k] =
    (J_trial - J_k + gam/s ||c_inc - c||^2_Q) / |J_k| for every trial looked at (> 0: rejected)."""
    from .fct import cost_functional, l2_norm_sq_Q
    n, Nt = nodes, num_steps
    tl = (Nt + 1) * n
    M = sb.cm.M
    uhat = np.asarray(uhat, dtype=np.float64)
    uk = np.zeros(tl)
    uk[:n] = u0
    if optim == "alltime":
        uk[n:] = uhat[n:]                       # uk = np.copy(uhat_all), level 0 = u0
    else:
        uk[Nt * n:] = uhat                      # uk[num_steps*nodes:] = uhat_T (finaltime.py:146)
    c_prev = np.array(c0, dtype=np.float64)
    pk = np.zeros(tl)
    hist = dict(cost=[], armijo_k=[], armijo_margin=[])
    for _ in range(iters):
        pk = solidbody_adjoint(sb, c_prev, uk, uhat, np.zeros(tl), n, Nt, dt, optim=optim)
        dk = solidbody_descent_direction(sb, c_prev, uk, pk, beta, n, Nt)
        ck = np.clip(c_prev + s0 * dk, c_lower, c_upper)
        solidbody_forward(sb, ck, uk, n, Nt, dt)
        J_k = cost_functional(uk, uhat, ck, Nt, dt, M, beta, optim)
        margins = []
        for k in range(max_armijo):
            s = s0 * (1 / 2 ** k)
            c_inc = np.clip(ck + s * dk, c_lower, c_upper)
            solidbody_forward(sb, c_inc, uk, n, Nt, dt)
            J = cost_functional(uk, uhat, c_inc, Nt, dt, M, beta, optim)
            stat = l2_norm_sq_Q(c_inc - ck, Nt, dt, M)
            margins.append((J - J_k + gam / s * stat) / abs(J_k))
            if not (J - J_k > -gam / s * stat):
                break
        hist["cost"].append(J)
        hist["armijo_k"].append(k + 1)
        hist["armijo_margin"].append(margins)
        c_prev = c_inc
    hist["armijo_margin_min"] = min(abs(m) for ms in hist["armijo_margin"] for m in ms)
    return uk, pk, c_prev, hist


# ---------------------------------------------------------------------------
# linear advection-diffusion with a distributed source control and a manufactured solution
#   advection_FCT_PDECO_alltime_exact.py (config C1's parameter set)
# ---------------------------------------------------------------------------
def exact_velocity(x, y):
    """advection_FCT_PDECO_alltime_exact.py:132-135 (unit square)."""
    return 2 * (y - 0.5) * x * (1 - x), -2 * (x - 0.5) * y * (1 - y)


class LinearSource:
    """Matrices of advection_FCT_PDECO_alltime_exact.py:160-176: A_u = A - eps*Ad, A_p = -A - eps*Ad."""

    def __init__(self, asm, eps=1e-3, wind=exact_velocity):
        self.asm = asm
        self.cm = _common(asm)
        self.eps = eps
        self.A = asm.convection(wind)
        self.A_u = self.A - eps * self.cm.Ad
        self.A_p = -self.A - eps * self.cm.Ad


def linear_forward(ls: LinearSource, src, uk, nodes, num_steps, dt):
    """:240-253: u_rhs = assemble((g_np1 + c_np1)*v*dx) = M (g + c)_{n+1}; FCT_alg(A_u, ...) (old sign).
    ``src`` = g + c as a trajectory."""
    uk[nodes:] = np.zeros(num_steps * nodes)
    for i in range(1, num_steps + 1):
        start, end = i * nodes, (i + 1) * nodes
        rhs = ls.cm.M @ src[start:end]
        uk[start:end] = ls.cm.fct(-ls.A_u, rhs, uk[start - nodes:start], dt)
    return uk


def linear_adjoint(ls: LinearSource, uk, uhat, pk, nodes, num_steps, dt):
    """:259-274: p(T) = 0, p_rhs = assemble((uhat_n - u_n)*v*dx), FCT_alg(A_p, ...)."""
    pk[:] = 0.0
    for i in reversed(range(0, num_steps)):
        start, end = i * nodes, (i + 1) * nodes
        rhs = ls.cm.M @ (uhat[start:end] - uk[start:end])
        pk[start:end] = ls.cm.fct(-ls.A_p, rhs, pk[end:end + nodes], dt)
    return pk


def exact_fields(t, X, Y, T=1.0, beta=1e-3, c_lower=0.0, c_upper=0.5, e1=0.2, e2=0.3, k1=1, k2=1, eps=1e-3):
    """uex, pex, cex, gex, uhatex of advection_FCT_PDECO_alltime_exact.py:77-128 at time t (inputs of the
    manufactured problem; X, Y = meshgrid of np.arange(a1, a2+dx, dx))."""
    pi = np.pi
    sx, sy, cx, cy = np.sin(k1 * pi * X), np.sin(k1 * pi * Y), np.cos(k1 * pi * X), np.cos(k1 * pi * Y)
    u = np.exp(e1 * t) * (sx * sy) ** 2
    amp = np.exp(e2 * T) - np.exp(e2 * t)
    sx2, sy2, cx2, cy2 = np.sin(k2 * pi * X), np.sin(k2 * pi * Y), np.cos(k2 * pi * X), np.cos(k2 * pi * Y)
    p = amp * (sx2 * sy2) ** 2
    c = np.clip(1 / beta * p, c_lower, c_upper)
    wx, wy = exact_velocity(X, Y)
    dudx = 2 * k1 * pi * np.exp(e1 * t) * sx * cx * sy ** 2
    dudy = 2 * k1 * pi * np.exp(e1 * t) * sx ** 2 * sy * cy
    du2dx2 = 2 * (pi * k1) ** 2 * np.exp(e1 * t) * np.cos(2 * k1 * pi * X) * sy ** 2
    du2dy2 = 2 * (pi * k1) ** 2 * np.exp(e1 * t) * sx ** 2 * np.cos(2 * k1 * pi * Y)
    g = e1 * u - eps * (du2dx2 + du2dy2) + wx * dudx + wy * dudy - c
    dpdt = -e2 * np.exp(e2 * t) * (sx2 * sy2) ** 2
    dpdx = 2 * k2 * pi * amp * sx2 * cx2 * sy2 ** 2
    dpdy = 2 * k2 * pi * amp * sx2 ** 2 * sy2 * cy2
    dp2dx2 = 2 * (pi * k2) ** 2 * amp * np.cos(2 * k2 * pi * X) * sy2 ** 2
    dp2dy2 = 2 * (pi * k2) ** 2 * amp * sx2 ** 2 * np.cos(2 * k2 * pi * Y)
    uhat = -dpdt - eps * (dp2dx2 + dp2dy2) - wx * dpdx - wy * dpdy + u
    return dict(u=u, p=p, c=c, g=g, uhat=uhat)
