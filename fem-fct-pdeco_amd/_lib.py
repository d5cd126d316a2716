"""ctypes binding of libfemfct.so (include/femfct.h).

The HIP library is the product: there is no CPU fallback.  Importing this module
raises ``ImportError`` with build instructions if the shared object is missing,
and every call raises ``FemFctError`` (``ValueError`` for FEMFCT_ERR_INVALID, as
the reference raises ``ValueError`` for invalid options, helpers.py:417-419).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FEMFCT_LIB: another build of the same library (the -DFEMFCT_TUNING build of `make tuning`); there is still no CPU path
LIB_PATH = os.environ.get("FEMFCT_LIB") or os.path.join(_HERE, "lib", "libfemfct.so")

OK, ERR_INVALID, ERR_HIP, ERR_NOT_CONVERGED, ERR_NOMEM = 0, 1, 2, 3, 4
FLAG_MMATRIX_ROWSUM, FLAG_SOLVER_BUDGET, FLAG_COARSE_ITERS, FLAG_CHEBYSHEV, FLAG_ROW_PAIRS = 1, 2, 4, 8, 16
ORDER_VERTEX, ORDER_FENICS = 0, 1
SOLVER_JACOBI, SOLVER_BICGSTAB = 0, 1
REGIME_ROWS, REGIME_STRIPS, REGIME_TILE32, REGIME_PATCH64, REGIME_MESH = 0, 1, 2, 3, 4
ABI_VERSION = 5


class FemFctError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfemfct error {code}: {msg}")
        self.code = code


class FemFctValueError(FemFctError, ValueError):
    pass


class NotConverged(FemFctError):
    pass


class StepInfo(C.Structure):
    _fields_ = [("flags", C.c_int32), ("solver_iters", C.c_int32),
                ("solver_resid", C.c_double), ("min_rowsum", C.c_double)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C fem-fct-pdeco_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    if lib.femfct_abi_version() != ABI_VERSION:
        raise ImportError("libfemfct.so ABI version mismatch: rebuild the library")
    return lib


lib = _load()

_p = C.c_void_p
_d = C.c_double
_i = C.c_int32
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# name -> (restype, argtypes); mirrors include/femfct.h one to one
SIGNATURES = {
    "femfct_abi_version": (C.c_int, []),
    "femfct_build_id": (C.c_char_p, []),
    "femfct_create": (C.c_int, [C.POINTER(_p), C.c_int]),
    "femfct_destroy": (C.c_int, [_p]),
    "femfct_last_error": (C.c_char_p, [_p]),
    "femfct_synchronize": (C.c_int, [_p]),
    "femfct_stream": (_p, [_p]),
    "femfct_set_solver": (C.c_int, [_p, C.c_int, _d, C.c_int]),
    "femfct_set_graphs": (C.c_int, [_p, C.c_int]),
    "femfct_graph_replay_active": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "femfct_rotation_derived": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "femfct_launch_info": (C.c_int, [_p, C.POINTER(C.c_int32)]),
    "femfct_set_fusion": (C.c_int, [_p, C.c_int, C.c_int]),
    "femfct_kernel_regime": (C.c_int, [_p, _i]),
    "femfct_patch_walkers": (C.c_int, [_p, _i, _i]),
    "femfct_lowop_nonzero_fraction": (C.c_int, [_p, _dp]),
    "femfct_set_profiling": (C.c_int, [_p, C.c_int]),
    "femfct_profile_report": (C.c_int, [_p, _p, _p, _i]),
    "femfct_malloc": (C.c_int, [_p, C.POINTER(_p), C.c_size_t]),
    "femfct_free": (C.c_int, [_p, _p]),
    "femfct_memcpy_h2d": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "femfct_memcpy_d2h": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "femfct_memcpy_d2d": (C.c_int, [_p, _p, _p, C.c_size_t]),
    "femfct_memset0": (C.c_int, [_p, _p, C.c_size_t]),
    "femfct_set_pattern_csr": (C.c_int, [_p, _i, _p, _p]),
    "femfct_set_mesh_square": (C.c_int, [_p, _d, _d, _i, _i]),
    "femfct_n": (_i, [_p]),
    "femfct_ell_width": (_i, [_p]),
    "femfct_get_ell_cols": (C.c_int, [_p, _p]),
    "femfct_csr_to_ell": (C.c_int, [_p, _p, _p]),
    "femfct_ell_to_csr": (C.c_int, [_p, _p, _p]),
    "femfct_set_mass": (C.c_int, [_p, _p, _p]),
    "femfct_mass_ell": (_p, [_p]),
    "femfct_stiffness_ell": (_p, [_p]),
    "femfct_lumped_mass": (_p, [_p]),
    "femfct_fct_step": (C.c_int, [_p, _p, _p, _i, _p, _p, _d, _p, _i]),
    "femfct_last_step_info": (C.c_int, [_p, C.POINTER(StepInfo), _i]),
    "femfct_fct_step_host": (C.c_int, [_p, _p, _p, _p, _p, _d, _p, C.POINTER(StepInfo)]),
    "femfct_chebsi": (C.c_int, [_p, _p, _p, _i, _d, _d, _i]),
    "femfct_chebsi_md": (C.c_int, [_p, _p, _p, _p, _i, _d, _d, _i]),
    "femfct_artificial_diffusion": (C.c_int, [_p, _p, _p, _i]),
    "femfct_spmv": (C.c_int, [_p, _p, _p, _d, _d, _p, _i]),
    "femfct_mesh_quad_points": (C.c_int, [_p, _p, _p]),
    "femfct_assemble_convection": (C.c_int, [_p, _p, _d, _p]),
    "femfct_assemble_rotation": (C.c_int, [_p, _d, _p]),
    "femfct_drift_gradient_rhs": (C.c_int, [_p, _p, _p, _p, _d, _d, _d, _p, _i]),
    "femfct_solidbody_forward": (C.c_int, [_p, _p, _p, _i, _p, _i, _d, _d, _d, _d, _d, _i]),
    "femfct_solidbody_forward_src": (C.c_int, [_p, _p, _p, _i, _p, _p, _i, _d, _d, _d, _d, _d, _i]),
    "femfct_solidbody_adjoint": (C.c_int, [_p, _p, _p, _i, _p, _p, _p, _i, _d, _d, _d, _d, _d, _i, _i]),
    "femfct_traj_info": (C.c_int, [_p, C.POINTER(StepInfo), _i, _i]),
    "femfct_descent_pointwise": (C.c_int, [_p, C.c_int64, _d, _p, _d, _p, _p, _d, _p]),
    "femfct_ell_transpose": (C.c_int, [_p, _p, _p]),
    "femfct_axpby": (C.c_int, [_p, C.c_int64, _d, _p, _d, _p, _p]),
    "femfct_set_krylov": (C.c_int, [_p, _d, _i]),
    "femfct_set_species_solver": (C.c_int, [_p, _i]),
    "femfct_bicgstab": (C.c_int, [_p, _p, _i, _p, _p, _p, _i, C.POINTER(StepInfo)]),
    "femfct_nonlinear_forward": (C.c_int, [_p, _p, _p, _p, _i, _d, _d, _i]),
    "femfct_nonlinear_adjoint": (C.c_int, [_p, _p, _p, _p, _p, _i, _d, _d, _i]),
    "femfct_schnak_forward": (C.c_int, [_p, _p, _p, _p, _p, _i, _d, _p, _d, _i]),
    "femfct_schnak_adjoint": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _d, _p, _i, _i]),
    "femfct_schnak_forward_tw": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _d, _p, _d, _i]),
    "femfct_schnak_adjoint_tw": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _d, _p, _i, _i]),
    "femfct_chtxs_forward": (C.c_int, [_p, _p, _p, _p, _i, _d, _p, _d, _i]),
    "femfct_chtxs_adjoint": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _d, _p, _d, _i, _i]),
    "femfct_traj_krylov_info": (C.c_int, [_p, C.POINTER(StepInfo), _i, _i]),
    "femfct_l2_norm_sq_Q": (C.c_int, [_p, _p, _p, _i, _d, _p, _i]),
    "femfct_l2_norm_sq_Omega": (C.c_int, [_p, _p, _p, _p, _i]),
    "femfct_cost_functional": (C.c_int, [_p, _p, _p, _p, _i, _i, _d, _d, _i, _p, _p, _p, _i]),
    "femfct_project_control": (C.c_int, [_p, _p, _d, _p, _d, _d, _p, C.c_int64]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def check(ctx_handle, code):
    if code == OK:
        return
    msg = lib.femfct_last_error(ctx_handle)
    msg = msg.decode() if msg else ""
    if code == ERR_INVALID:
        raise FemFctValueError(code, msg)
    if code == ERR_NOT_CONVERGED:
        raise NotConverged(code, msg)
    raise FemFctError(code, msg)
