"""Loading of the committed golden vectors (tests/golden/*.npz)."""
import os

import numpy as np
from scipy.sparse import csr_matrix, diags

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def csr_from(z, prefix, n, key=""):
    return csr_matrix((z[f"{key}{prefix}_data"], z[f"{key}{prefix}_indices"], z[f"{key}{prefix}_indptr"]), shape=(n, n))


def fct_case(z, name):
    k = name + "/"
    a1, a2, nc = z[k + "geom"]
    n = (int(nc) + 1) ** 2
    c = dict(name=name, a1=float(a1), a2=float(a2), n_cells=int(nc), n=n,
             A=csr_from(z, "A", n, k), M=csr_from(z, "M", n, k), ml=z[k + "ml"],
             rhs=z[k + "rhs"], u_n=z[k + "u_n"], dt=float(z[k + "dt"]),
             N=csr_from(z, "N", n, k) if int(z[k + "has_nfm"]) else None,
             u_np1=z[k + "u_np1"], mmatrix_failed=bool(z[k + "mmatrix_failed"]))
    c["ML"] = diags(c["ml"]).tocsr()
    return c


def fct_case_names():
    z = load("fct_cases.npz")
    return [str(s) for s in z["names"]]
