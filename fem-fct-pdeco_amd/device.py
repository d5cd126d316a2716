"""Thin object layer over the C ABI: one ``Context`` = one femfct_ctx (one GPU,
one HIP stream), ``DeviceArray`` = a float64 device buffer.

Any object exposing ``data_ptr()`` (e.g. a CUDA/ROCm ``torch.Tensor`` of dtype
float64) is accepted wherever a device array is expected, so callers that
already hold HBM-resident tensors hand them over without a copy.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check, StepInfo


def _host_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _as_f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceArray:
    """float64 buffer in HBM owned by a Context."""

    def __init__(self, ctx: "Context", count: int):
        self.ctx = ctx
        self.count = int(count)
        ptr = C.c_void_p()
        check(ctx.handle, lib.femfct_malloc(ctx.handle, C.byref(ptr), self.count * 8))
        self.ptr = ptr.value
        ctx._arrays.add(self)

    def data_ptr(self) -> int:
        return self.ptr

    def upload(self, host) -> "DeviceArray":
        h = _as_f64(host).reshape(-1)
        if h.size != self.count:
            raise ValueError(f"upload: host array has {h.size} elements, device array {self.count}")
        check(self.ctx.handle, lib.femfct_memcpy_h2d(self.ctx.handle, self.ptr, _host_ptr(h), h.nbytes))
        return self

    def download(self, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.empty(self.count, dtype=np.float64)
        if out.size != self.count or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("download: need a C-contiguous float64 array of matching size")
        check(self.ctx.handle, lib.femfct_memcpy_d2h(self.ctx.handle, _host_ptr(out), self.ptr, out.nbytes))
        return out

    def zero(self) -> "DeviceArray":
        check(self.ctx.handle, lib.femfct_memset0(self.ctx.handle, self.ptr, self.count * 8))
        return self

    def copy_from(self, other, count=None, dst_off=0, src_off=0) -> "DeviceArray":
        count = self.count if count is None else int(count)
        check(self.ctx.handle, lib.femfct_memcpy_d2d(self.ctx.handle, self.ptr + 8 * dst_off,
                                                      dptr(other) + 8 * src_off, count * 8))
        return self

    def free(self):
        if self.ptr:
            lib.femfct_free(self.ctx.handle, self.ptr)
            self.ptr = 0
            self.ctx._arrays.discard(self)


def _wind_scale(wind_scale, num_steps) -> np.ndarray:
    ws = _as_f64(wind_scale).reshape(-1)
    if ws.size != num_steps + 1:
        raise ValueError(f"wind_scale: {ws.size} values, expected num_steps + 1 = {num_steps + 1} (s(t_0) .. s(t_Nt))")
    return ws


def dptr(x) -> int:
    """device address of a DeviceArray / torch tensor / raw int (None -> 0)."""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    return int(x.data_ptr())


class Context:
    """femfct_ctx wrapper.  ``Context.n`` / ``.W`` describe the registered pattern."""

    def __init__(self, device_id: int = 0):
        h = C.c_void_p()
        code = lib.femfct_create(C.byref(h), int(device_id))
        if code != _lib.OK:
            raise _lib.FemFctError(code, f"femfct_create(device {device_id}) failed: no usable HIP device")
        self.handle = h
        self.device_id = int(device_id)
        self._arrays = set()
        self.mesh = None

    # -- lifetime ----------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None):
            for a in list(self._arrays):
                a.free()
            lib.femfct_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- settings ----------------------------------------------------------
    def set_solver(self, solver=_lib.SOLVER_JACOBI, rel_tol=1e-13, max_iters=400):
        check(self.handle, lib.femfct_set_solver(self.handle, solver, rel_tol, max_iters))

    def set_graphs(self, enable: bool):
        check(self.handle, lib.femfct_set_graphs(self.handle, int(bool(enable))))

    def launch_info(self) -> dict:
        """Bandwidth-regime kernels of the most recent step / sweep (diagnostic): Jacobi launch kind, its walkers, interior
        patches per side of the split Chebyshev launch, halo depth."""
        out = (C.c_int32 * 4)()
        check(self.handle, lib.femfct_launch_info(self.handle, out))
        kinds = {0: "other", 1: "k_strip4_jacobi", 2: "k_strip4_jacobi_walk", 3: "k_strip_jacobi_pair_walk"}
        return {"jacobi_kernel": kinds.get(out[0], "?"), "jacobi_walkers": int(out[1]), "cheb_interior_patches": int(out[2]),
                "halo": int(out[3])}

    def graph_replay_active(self) -> bool:
        """False while sweeps are enqueued kernel by kernel: graphs off, per-class profiling, or rocprofv3 attached."""
        v = C.c_int(0)
        check(self.handle, lib.femfct_graph_replay_active(self.handle, C.byref(v)))
        return bool(v.value)

    def rotation_derived(self) -> bool:
        """True when the most recent solid-body sweep evaluated the rotation operator from the node positions."""
        v = C.c_int(0)
        check(self.handle, lib.femfct_rotation_derived(self.handle, C.byref(v)))
        return bool(v.value)

    KERNEL_CLASSES = ("build_low", "jacobi", "dudt_rhs", "cheb", "flux", "limit", "assemble", "other")

    def set_profiling(self, enable: bool):
        check(self.handle, lib.femfct_set_profiling(self.handle, int(bool(enable))))

    def profile_report(self):
        """{class: (total_ms, launches)} of the kernels run since profiling was enabled / last report."""
        ms = np.zeros(8)
        cnt = np.zeros(8, dtype=np.int32)
        check(self.handle, lib.femfct_profile_report(self.handle, _host_ptr(ms), _host_ptr(cnt), 8))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KERNEL_CLASSES)}

    def set_fusion(self, strips=True, tiles=True):
        check(self.handle, lib.femfct_set_fusion(self.handle, int(bool(strips)), int(bool(tiles))))

    def patch_walkers(self, batch=1, sweeps=36) -> int:
        """Persistent workgroups per batch member of the bandwidth-regime Jacobi / Chebyshev launches (0: one
        workgroup per patch)."""
        return lib.femfct_patch_walkers(self.handle, int(batch), int(sweeps))

    def kernel_regime(self, batch=1) -> int:
        """_lib.REGIME_*: the Jacobi / Chebyshev kernel family a step with ``batch`` members runs."""
        return lib.femfct_kernel_regime(self.handle, int(batch))

    def lowop_nonzero_fraction(self) -> float:
        """share of the low-order operator's off-diagonals the bandwidth-regime Jacobi launches load (1.0: all)"""
        out = C.c_double(1.0)
        check(self.handle, lib.femfct_lowop_nonzero_fraction(self.handle, C.byref(out)))
        return float(out.value)

    def uses_bandwidth_tiles(self, batch=1) -> bool:
        return self.kernel_regime(batch) == _lib.REGIME_PATCH64

    def synchronize(self):
        check(self.handle, lib.femfct_synchronize(self.handle))

    @property
    def stream(self) -> int:
        return lib.femfct_stream(self.handle) or 0

    # -- memory --------------------------------------------------------------
    def empty(self, count) -> DeviceArray:
        return DeviceArray(self, count)

    def zeros(self, count) -> DeviceArray:
        return DeviceArray(self, count).zero()

    def array(self, host) -> DeviceArray:
        h = _as_f64(host).reshape(-1)
        return DeviceArray(self, h.size).upload(h)

    # -- pattern / constant operators -----------------------------------------
    @property
    def n(self) -> int:
        return lib.femfct_n(self.handle)

    @property
    def W(self) -> int:
        return lib.femfct_ell_width(self.handle)

    def set_pattern_csr(self, indptr, indices):
        ip = np.ascontiguousarray(indptr, dtype=np.int32)
        ix = np.ascontiguousarray(indices, dtype=np.int32)
        check(self.handle, lib.femfct_set_pattern_csr(self.handle, ip.size - 1, _host_ptr(ip), _host_ptr(ix)))
        self.mesh = None

    def set_mesh_square(self, a1, a2, n_cells, order=_lib.ORDER_FENICS):
        check(self.handle, lib.femfct_set_mesh_square(self.handle, float(a1), float(a2), int(n_cells), int(order)))

    def ell_cols(self) -> np.ndarray:
        out = np.empty(self.W * self.n, dtype=np.int32)
        check(self.handle, lib.femfct_get_ell_cols(self.handle, _host_ptr(out)))
        return out.reshape(self.W, self.n)

    def set_mass(self, M_csr_vals, ml):
        m = _as_f64(M_csr_vals)
        l = _as_f64(ml)
        check(self.handle, lib.femfct_set_mass(self.handle, _host_ptr(m), _host_ptr(l)))

    def csr_to_ell(self, csr_vals, out: DeviceArray | None = None) -> DeviceArray:
        v = _as_f64(csr_vals)
        if out is None:
            out = self.empty(self.W * self.n)
        check(self.handle, lib.femfct_csr_to_ell(self.handle, _host_ptr(v), dptr(out)))
        return out

    def ell_to_csr(self, ell, nnz: int) -> np.ndarray:
        out = np.empty(int(nnz), dtype=np.float64)
        check(self.handle, lib.femfct_ell_to_csr(self.handle, dptr(ell), _host_ptr(out)))
        return out

    @property
    def mass_ell(self) -> int:
        return lib.femfct_mass_ell(self.handle) or 0

    @property
    def stiffness_ell(self) -> int:
        return lib.femfct_stiffness_ell(self.handle) or 0

    @property
    def lumped_mass(self) -> int:
        return lib.femfct_lumped_mass(self.handle) or 0

    # -- step operator -----------------------------------------------------------
    def fct_step(self, A_ell, u_n, dt, u_out, rhs=None, N_ell=None, N_shared=False, batch=1):
        check(self.handle, lib.femfct_fct_step(self.handle, dptr(A_ell), dptr(N_ell), int(bool(N_shared)),
                                               dptr(rhs), dptr(u_n), float(dt), dptr(u_out), int(batch)))

    def last_step_info(self, batch=1):
        arr = (StepInfo * batch)()
        check(self.handle, lib.femfct_last_step_info(self.handle, arr, int(batch)))
        return [dict(flags=a.flags, solver_iters=a.solver_iters, solver_resid=a.solver_resid,
                     min_rowsum=a.min_rowsum) for a in arr]

    def fct_step_host(self, A_csr_vals, rhs, u_n, dt, N_csr_vals=None):
        a = _as_f64(A_csr_vals)
        u = _as_f64(u_n)
        r = None if rhs is None else _as_f64(rhs)
        nn = None if N_csr_vals is None else _as_f64(N_csr_vals)
        out = np.empty(self.n, dtype=np.float64)
        info = StepInfo()
        check(self.handle, lib.femfct_fct_step_host(
            self.handle, _host_ptr(a), None if nn is None else _host_ptr(nn),
            None if r is None else _host_ptr(r), _host_ptr(u), float(dt), _host_ptr(out), C.byref(info)))
        return out, dict(flags=info.flags, solver_iters=info.solver_iters, solver_resid=info.solver_resid,
                         min_rowsum=info.min_rowsum)

    def chebsi(self, b, y, cheb_iter=20, lmin=0.5, lmax=2.0, batch=1):
        check(self.handle, lib.femfct_chebsi(self.handle, dptr(b), dptr(y), int(cheb_iter), float(lmin),
                                             float(lmax), int(batch)))

    def chebsi_md(self, b, y, md, cheb_iter=20, lmin=0.5, lmax=2.0):
        """ChebSI with a preconditioner diagonal of the caller's (not diag(M)); one system."""
        check(self.handle, lib.femfct_chebsi_md(self.handle, dptr(b), dptr(y), dptr(md), int(cheb_iter), float(lmin),
                                                float(lmax), 1))

    def artificial_diffusion(self, K_ell, D_ell, batch=1):
        check(self.handle, lib.femfct_artificial_diffusion(self.handle, dptr(K_ell), dptr(D_ell), int(batch)))

    def spmv(self, mat_ell, x, y, alpha=1.0, beta=0.0, batch=1):
        check(self.handle, lib.femfct_spmv(self.handle, dptr(mat_ell), dptr(x), float(alpha), float(beta),
                                           dptr(y), int(batch)))

    # -- structured assembly --------------------------------------------------------
    def quad_points(self, n_cells):
        cnt = n_cells * n_cells * 2 * 6
        xq = np.empty(cnt)
        yq = np.empty(cnt)
        check(self.handle, lib.femfct_mesh_quad_points(self.handle, _host_ptr(xq), _host_ptr(yq)))
        return xq, yq

    def assemble_convection(self, wind_q, scale=1.0, out: DeviceArray | None = None) -> DeviceArray:
        w = _as_f64(wind_q)
        if out is None:
            out = self.empty(self.W * self.n)
        check(self.handle, lib.femfct_assemble_convection(self.handle, _host_ptr(w), float(scale), dptr(out)))
        return out

    def assemble_rotation(self, omega, out: DeviceArray | None = None) -> DeviceArray:
        """``assemble_sparse(dot(wind, grad(v))*u*dx)`` for ``wind = omega * (-x[1], x[0])`` in closed form (equal to
        :meth:`assemble_convection` of that wind up to rounding); the large-mesh step kernels recognise it and derive
        its rows instead of loading them."""
        if out is None:
            out = self.empty(self.W * self.n)
        check(self.handle, lib.femfct_assemble_rotation(self.handle, float(omega), dptr(out)))
        return out

    def drift_gradient_rhs(self, c, u, p, beta, out, levels, drift=(1.0, 1.0)):
        check(self.handle, lib.femfct_drift_gradient_rhs(self.handle, dptr(c), dptr(u), dptr(p), float(beta),
                                                         float(drift[0]), float(drift[1]), dptr(out), int(levels)))

    # -- trajectories ------------------------------------------------------------------
    def solidbody_forward(self, Arot, c_traj, u_traj, num_steps, dt, eps=0.0, rot_scale=1.0,
                          drift=(1.0, 1.0), batch=1, c_shared=False, src_traj=None):
        check(self.handle, lib.femfct_solidbody_forward_src(
            self.handle, dptr(Arot), dptr(c_traj), int(bool(c_shared)), dptr(src_traj), dptr(u_traj), int(num_steps),
            float(dt), float(eps), float(rot_scale), float(drift[0]), float(drift[1]), int(batch)))

    def solidbody_adjoint(self, Arot, c_traj, u_traj, uhat, p_traj, num_steps, dt, eps=0.0, rot_scale=1.0,
                          drift=(1.0, 1.0), alltime=False, batch=1, c_shared=False):
        check(self.handle, lib.femfct_solidbody_adjoint(
            self.handle, dptr(Arot), dptr(c_traj), int(bool(c_shared)), dptr(u_traj), dptr(uhat), dptr(p_traj),
            int(num_steps), float(dt), float(eps), float(rot_scale), float(drift[0]), float(drift[1]),
            int(bool(alltime)), int(batch)))

    def traj_info(self, num_steps, batch=1):
        arr = (StepInfo * (num_steps * batch))()
        check(self.handle, lib.femfct_traj_info(self.handle, arr, int(num_steps), int(batch)))
        return dict(flags=np.array([a.flags for a in arr]).reshape(num_steps, batch),
                    solver_iters=np.array([a.solver_iters for a in arr]).reshape(num_steps, batch),
                    solver_resid=np.array([a.solver_resid for a in arr]).reshape(num_steps, batch),
                    min_rowsum=np.array([a.min_rowsum for a in arr]).reshape(num_steps, batch))

    # -- optimisation layer ------------------------------------------------------------
    def l2_norm_sq_Q(self, a, b, num_steps, dt, batch=1) -> np.ndarray:
        out = np.empty(batch)
        check(self.handle, lib.femfct_l2_norm_sq_Q(self.handle, dptr(a), dptr(b), int(num_steps), float(dt),
                                                   _host_ptr(out), int(batch)))
        return out

    def l2_norm_sq_Omega(self, a, b, batch=1) -> np.ndarray:
        out = np.empty(batch)
        check(self.handle, lib.femfct_l2_norm_sq_Omega(self.handle, dptr(a), dptr(b), _host_ptr(out), int(batch)))
        return out

    def cost_functional(self, var1, var1_target, control, num_steps, dt, beta, optim, var2=None,
                        var2_target=None, batch=1) -> np.ndarray:
        if optim not in ("alltime", "finaltime"):
            raise ValueError(f"Invalid value for 'optim': '{optim}'. Must be one of ['alltime', 'finaltime'].")
        out = np.empty(batch)
        check(self.handle, lib.femfct_cost_functional(
            self.handle, dptr(var1), dptr(var1_target), dptr(control), 0, int(num_steps), float(dt), float(beta),
            int(optim == "finaltime"), dptr(var2), dptr(var2_target), _host_ptr(out), int(batch)))
        return out

    def project_control(self, c, s, d, c_lower, c_upper, out, count):
        check(self.handle, lib.femfct_project_control(self.handle, dptr(c), float(s), dptr(d), float(c_lower),
                                                      float(c_upper), dptr(out), int(count)))

    # -- non-FCT species / PDE systems ---------------------------------------------------
    def descent_pointwise(self, count, beta, c, x, out, y=None, scale=1.0, divisor=1.0):
        """out = -(beta*c - t), t = x*y/divisor (y given) or scale*x"""
        check(self.handle, lib.femfct_descent_pointwise(self.handle, int(count), float(beta), dptr(c), float(scale), dptr(x),
                                                       dptr(y), float(divisor), dptr(out)))

    def ell_transpose(self, src, out=None) -> DeviceArray:
        if out is None:
            out = self.empty(self.W * self.n)
        check(self.handle, lib.femfct_ell_transpose(self.handle, dptr(src), dptr(out)))
        return out

    def axpby(self, count, alpha, a, beta, b, out):
        check(self.handle, lib.femfct_axpby(self.handle, int(count), float(alpha), dptr(a), float(beta), dptr(b), dptr(out)))

    def set_krylov(self, rel_tol=1e-13, max_iters=2000):
        check(self.handle, lib.femfct_set_krylov(self.handle, float(rel_tol), int(max_iters)))

    def set_species_solver(self, mode="auto"):
        """'auto': tile-fused Chebyshev for the non-FCT solves of the sweeps where it applies; 'bicgstab'."""
        check(self.handle, lib.femfct_set_species_solver(self.handle, {"auto": 0, "bicgstab": 1}[mode]))

    def bicgstab(self, mat_ell, b, x0, x, batch=1, mat_shared=False):
        arr = (StepInfo * batch)()
        check(self.handle, lib.femfct_bicgstab(self.handle, dptr(mat_ell), int(bool(mat_shared)), dptr(b), dptr(x0),
                                               dptr(x), int(batch), arr))
        return [dict(flags=a.flags, solver_iters=a.solver_iters, solver_resid=a.solver_resid) for a in arr]

    def nonlinear_forward(self, Aw, c_level, u, num_steps, dt, eps, batch=1):
        check(self.handle, lib.femfct_nonlinear_forward(self.handle, dptr(Aw), dptr(c_level), dptr(u), int(num_steps),
                                                        float(dt), float(eps), int(batch)))

    def nonlinear_adjoint(self, Aw, u, uhat_T, p, num_steps, dt, eps, batch=1):
        check(self.handle, lib.femfct_nonlinear_adjoint(self.handle, dptr(Aw), dptr(u), dptr(uhat_T), dptr(p),
                                                        int(num_steps), float(dt), float(eps), int(batch)))

    def schnak_forward(self, Aw, c_level, u, v, num_steps, dt, par, rescaling=1.0, batch=1, wind_scale=None):
        """wind_scale: None (stationary wind) or the num_steps+1 factors s(t_k) of a separable wind s(t) w0(x)."""
        par = _as_f64(par)
        ws = None if wind_scale is None else _wind_scale(wind_scale, num_steps)
        check(self.handle, lib.femfct_schnak_forward_tw(self.handle, dptr(Aw), None if ws is None else _host_ptr(ws),
                                                        dptr(c_level), dptr(u), dptr(v), int(num_steps), float(dt),
                                                        _host_ptr(par), float(rescaling), int(batch)))

    def schnak_adjoint(self, AwT, u, v, uhat_T, vhat_T, p, q, num_steps, dt, par, batch=1, alltime=False, wind_scale=None):
        par = _as_f64(par)
        ws = None if wind_scale is None else _wind_scale(wind_scale, num_steps)
        check(self.handle, lib.femfct_schnak_adjoint_tw(self.handle, dptr(AwT), None if ws is None else _host_ptr(ws),
                                                        dptr(u), dptr(v), dptr(uhat_T), dptr(vhat_T), dptr(p), dptr(q),
                                                        int(num_steps), float(dt), _host_ptr(par), int(bool(alltime)),
                                                        int(batch)))

    def chtxs_forward(self, c_level, u, v, num_steps, dt, par, rescaling=0.1, batch=1):
        par = _as_f64(par)
        check(self.handle, lib.femfct_chtxs_forward(self.handle, dptr(c_level), dptr(u), dptr(v), int(num_steps),
                                                    float(dt), _host_ptr(par), float(rescaling), int(batch)))

    def chtxs_adjoint(self, u, v, uhat, vhat, p, q, c, num_steps, dt, par, rescaling=0.1, alltime=True, batch=1):
        par = _as_f64(par)
        check(self.handle, lib.femfct_chtxs_adjoint(self.handle, dptr(u), dptr(v), dptr(uhat), dptr(vhat), dptr(p), dptr(q),
                                                    dptr(c), int(num_steps), float(dt), _host_ptr(par), float(rescaling),
                                                    int(bool(alltime)), int(batch)))

    def traj_krylov_info(self, num_steps, batch=1):
        arr = (StepInfo * (num_steps * batch))()
        check(self.handle, lib.femfct_traj_krylov_info(self.handle, arr, int(num_steps), int(batch)))
        return dict(flags=np.array([a.flags for a in arr]).reshape(num_steps, batch),
                    solver_iters=np.array([a.solver_iters for a in arr]).reshape(num_steps, batch),
                    solver_resid=np.array([a.solver_resid for a in arr]).reshape(num_steps, batch))
