"""Shared set-up of the example drivers (initial conditions and paths of the reference's scripts)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hp = importlib.import_module("fem-fct-pdeco_amd")
solvers = importlib.import_module("fem-fct-pdeco_amd.solvers")
pdeco = importlib.import_module("fem-fct-pdeco_amd.pdeco")
sweep = importlib.import_module("fem-fct-pdeco_amd.sweep")
hp.fct_helpers.VERBOSE = False


def grid(a1, a2, dx):
    """the np.arange grid every script samples its initial condition on (vertex order)"""
    X = np.arange(a1, a2 + dx, dx)
    return np.meshgrid(X, X)


def slotted_disc(a1, a2, dx, slit=0.05):
    """advection_solidbody_FCT_PDECO_finaltime.py:71-88"""
    X, Y = grid(a1, a2, dx)
    R = np.sqrt(X ** 2 + (Y - 1 / 3) ** 2)
    return ((R < 1 / 3) & ((np.abs(X) > slit) | (Y > 0.5))).astype(np.float64).reshape(-1)


def gaussian(a1, a2, dx):
    """advection_solidbody_FCT_PDECO_alltime.py:93-96"""
    X, Y = grid(a1, a2, dx)
    return np.exp(-20 * ((X + 2 / 3) ** 2 + 5 * (Y + 5 / 6) ** 2)).reshape(-1)


def to_dof(mesh, v):
    return hp.reorder_vector_to_dof(v, v.size // mesh.nodes, mesh.nodes, mesh.vertex_to_dof)
