"""femfct-mi355x: FEM-FCT forward/adjoint time-stepping core for MI355X (gfx950).

Drop-in for the FCT operator API of KarolinaBenkova/FEM-FCT-PDECO's ``helpers.py``:
hand-written HIP kernels behind a C ABI (``include/femfct.h``), bound with ctypes.
Importing this package loads ``lib/libfemfct.so`` and fails loudly if it is missing --
there is no CPU fallback.

    import importlib; hp = importlib.import_module("fem-fct-pdeco_amd")   # or: import femfct_amd as hp
    u_np1 = hp.FCT_alg_ref(A, rhs, u_n, dt, nodes, M, M_lumped, dof_neighbors)
"""
from ._lib import (FemFctError, FemFctValueError, NotConverged, LIB_PATH,  # noqa: F401
                   ORDER_VERTEX, ORDER_FENICS, SOLVER_JACOBI, SOLVER_BICGSTAB,
                   FLAG_MMATRIX_ROWSUM, FLAG_SOLVER_BUDGET, FLAG_COARSE_ITERS, FLAG_CHEBYSHEV, FLAG_ROW_PAIRS)
from .device import Context, DeviceArray  # noqa: F401
from .mesh import SquareMeshP1  # noqa: F401
from .fct_helpers import (  # noqa: F401
    FCT_alg_ref, FCT_alg, ChebSI, artificial_diffusion_mat, row_lump, sparse_nonzero, rel_err,
    find_node_neighbours, L2_norm_sq_Q, L2_norm_sq_Omega, cost_functional,
    reorder_vector_to_dof, reorder_vector_from_dof, reorder_vector_to_dof_time,
    reorder_vector_from_dof_time, set_device)
from .systems import (  # noqa: F401
    solve_nonlinear_equation, solve_adjoint_nonlinear_equation, solve_schnak_system,
    solve_adjoint_schnak_system, solve_chtxs_system, solve_adjoint_chtxs_system,
    get_schnak_sys_params, get_nonlinear_eqns_params, get_chtxs_sys_params,
    schnak_sys_IC, nonlinear_equation_IC, chtxs_sys_IC, armijo_line_search_ref, assemble_mass)
from .data_io import (import_data_final, extract_data, save_trajectory, save_results,  # noqa: F401
                      append_results_ledger)
from .pdeco import projected_gradient_descent, SystemPDECO  # noqa: F401
from . import fct_helpers, systems, solvers, sweep, data_io, pdeco  # noqa: F401

__version__ = "0.1.0"
