"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/femfct.h declares; host-side argument validation needs no GPU."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "femfct.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(femfct_[a-z0-9_A-Z]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    pkg = importlib.import_module("fem-fct-pdeco_amd")
    lib = ctypes.CDLL(pkg.LIB_PATH)
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    hdr = open(os.path.join(ROOT, "include", "femfct.h")).read()
    assert lib.femfct_abi_version() == int(re.search(r"#define\s+FEMFCT_ABI_VERSION\s+(\d+)", hdr).group(1)) == 5
    assert lib.femfct_abi_version() == importlib.import_module("fem-fct-pdeco_amd._lib").ABI_VERSION
    lib.femfct_build_id.restype = ctypes.c_char_p
    assert re.fullmatch(r"[0-9a-f]{16}", lib.femfct_build_id().decode())


def test_python_binding_covers_header():
    _lib = importlib.import_module("fem-fct-pdeco_amd._lib")
    assert sorted(_lib.SIGNATURES) == _declared()


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = importlib.import_module("fem-fct-pdeco_amd")
    with pytest.raises(pkg.FemFctError):
        pkg.Context(0)


def test_host_mesh_descriptor_matches_oracle():
    pkg = importlib.import_module("fem-fct-pdeco_amd")
    from oracle.mesh import SquareMesh
    for nc in (1, 4, 10):
        m = pkg.SquareMeshP1(-1, 1, nc)
        o = SquareMesh(-1, 1, nc)
        assert np.array_equal(m.vertex_to_dof, o.vertex_to_dof)
        assert [sorted(a) for a in m.dof_neighbors()] == [sorted(a) for a in o.dof_neighbors()]
        assert [a[-1] for a in m.dof_neighbors()] == list(range(m.nodes))  # self last (helpers.py:298)
    v = np.arange(2 * 25, dtype=float)
    m = pkg.SquareMeshP1(0, 1, 4)
    from oracle.mesh import reorder_vector_to_dof, reorder_vector_from_dof
    assert np.array_equal(pkg.reorder_vector_to_dof(v, 2, 25, m.vertex_to_dof), reorder_vector_to_dof(v, 2, 25, m.vertex_to_dof))
    assert np.array_equal(pkg.reorder_vector_from_dof_time(v, 2, 25, m.vertex_to_dof), reorder_vector_from_dof(v, 2, 25, m.vertex_to_dof))
