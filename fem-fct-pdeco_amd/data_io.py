"""Host I/O compatibility with the reference's data files: single-line,
comma-separated float trajectories written with ``ndarray.tofile(sep=',')`` and read with
``np.genfromtxt`` (helpers.py:1874-1956; Schnak_FCT_PDECO_refactored.py:271-275).  Pure host code:
it lets a user of the reference keep their post-processing when the solver backend is swapped."""
from __future__ import annotations

import os

import numpy as np

from .mesh import reorder_vector_from_dof


def import_data_final(file_path, nodes, vertex_to_dof, num_steps=0, time_dep=False):
    """helpers.py:1874-1911: returns ``(data_re, data)`` -- vertex-ordered (2-D for one frame) and
    DoF-ordered views of a stored trajectory."""
    sqnodes = round(np.sqrt(nodes))
    data = np.genfromtxt(file_path, delimiter=",")
    if time_dep:
        data = data[:(num_steps + 1) * nodes]
        data_re = reorder_vector_from_dof(data, num_steps + 1, nodes, vertex_to_dof)
    else:
        data = data[num_steps * nodes:(num_steps + 1) * nodes]
        data_re = reorder_vector_from_dof(data, 1, nodes, vertex_to_dof).reshape((sqnodes, sqnodes))
    return data_re, data


def extract_data(file_path, file_name, T, dt, nodes, vertex_to_dof=None):
    """helpers.py:1913-1956: cut the time level ``round(T/dt)`` out of ``<file_name>.csv`` and save it
    as ``<file_name>_T<T>.csv`` (one value per line, like ``np.savetxt``)."""
    idx = round(T / dt)
    input_file = os.path.join(file_path, f"{file_name}.csv")
    output_file = os.path.join(file_path, f"{file_name}_T{T}.csv")
    row = np.genfromtxt(input_file, delimiter=",")
    np.savetxt(output_file, row[idx * nodes:(idx + 1) * nodes], delimiter=",")
    print(f"Extracted data at {T=} into {output_file}.")
    return None


def save_trajectory(path, vec):
    """``vec.tofile(path, sep=',')`` (advection_solidbody_FCT_PDECO_finaltime.py:269-271)."""
    np.asarray(vec, dtype=np.float64).tofile(path, sep=",")
