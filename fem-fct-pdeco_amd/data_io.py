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
    as ``<file_name>_T<T>.csv`` (one value per line, ``np.savetxt``).  Like the reference it parses the
    columns with ``pandas.read_csv`` -- whose default float parser is not the correctly rounded one of
    ``np.genfromtxt``: the extracted file can differ from the stored numbers in the last digit, and a
    drop-in has to write the same bytes (tests/test_data_io.py pins them to a reference-written file)."""
    import pandas as pd
    idx = round(T / dt)
    start_col, end_col = idx * nodes, (idx + 1) * nodes
    input_file = os.path.join(file_path, f"{file_name}.csv")
    output_file = os.path.join(file_path, f"{file_name}_T{T}.csv")
    data = pd.read_csv(input_file, header=None, usecols=range(start_col, end_col), nrows=1)
    np.savetxt(output_file, data.to_numpy().flatten(), delimiter=",")
    print(f"Extracted data at {T=} into {output_file}.")
    return None


def save_trajectory(path, vec):
    """``vec.tofile(path, sep=',')`` (advection_solidbody_FCT_PDECO_finaltime.py:269-271)."""
    np.asarray(vec, dtype=np.float64).tofile(path, sep=",")


# ----------------------------------------------------------------------------- results of a PDECO run
# field lists of the three refactored drivers' "simulation results" ledgers
LEDGER_FIELDS = {
    # nonlinear_FCT_PDECO_refactored.py:242-256
    "nonlinear": ["timestamp", "Sim. duration", "T", "T_data", "beta", "tol", "GD its", "Armijo its", "C_ad",
                  "Mean c. in L^2(Q)^2", "Misfit norm", "J(c_true)", "out_folder_name"],
    # Schnak_FCT_PDECO_refactored.py:278-292
    "schnak": ["timestamp", "Sim. duration", "T", "T_data", "beta", "tol", "GD its", "Armijo its", "C_ad",
               "Mean c. in L^2(Q)^2", "Misfit norm u", "Misfit norm v", "J(c_true)", "out_folder_name"],
    # chemotaxis_FCT_PDECO_AT_refactored.py:285-302
    "chtxs": ["timestamp", "Sim. duration", "T", "beta", "tol", "GD its", "Armijo its", "C_ad", "Mean c. in L^2(Q)^2",
              "Misfit norm u", "Misfit norm v", "J(c_true)", "J_final_it", "J_diff", "out_folder_name"],
}
LEDGER_FILES = {"nonlinear": "NL_FT_simulation_results.csv", "schnak": "AdvSchnak_FT_simulation_results.csv",
                "chtxs": "Chtx_AT_simulation_results.csv"}
RESULT_PREFIX = {"nonlinear": "NL", "schnak": "AdvSchnak", "chtxs": "Chtx"}


def save_results(out_folder, problem, **arrays):
    """``uk.tofile(out_folder + "/AdvSchnak_u.csv", sep=",")`` for every given array
    (Schnak_FCT_PDECO_refactored.py:271-275, nonlinear_FCT_PDECO_refactored.py:237-239): one single-line CSV per
    trajectory, named ``<prefix>_<key>.csv`` with the drivers' prefixes.  Returns the written paths."""
    os.makedirs(out_folder, exist_ok=True)
    prefix = RESULT_PREFIX.get(problem, problem)
    paths = {}
    for key, vec in arrays.items():
        paths[key] = os.path.join(out_folder, f"{prefix}_{key}.csv")
        save_trajectory(paths[key], vec)
    return paths


def append_results_ledger(problem, row: dict, csv_file_path=None, timestamp=None):
    """Append one run to the driver's results ledger (Schnak_FCT_PDECO_refactored.py:277-299 and the two sibling
    scripts): same file name, column names and order, header written only when the file is new, ``csv.DictWriter``
    in append mode with ``newline=""``.  ``row`` holds every column but the timestamp; ``C_ad`` may be given as the
    pair ``(c_lower, c_upper)`` and is formatted ``"[lo, hi]"`` like the scripts do; ``Sim. duration`` is rounded to
    two decimals.  Unknown or missing columns raise ``ValueError`` (DictWriter's own behaviour for extras)."""
    import csv
    from datetime import datetime
    if problem not in LEDGER_FIELDS:
        raise ValueError(f"unknown problem '{problem}' (one of {sorted(LEDGER_FIELDS)})")
    fieldnames = LEDGER_FIELDS[problem]
    data = dict(row)
    data["timestamp"] = timestamp or datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    if isinstance(data.get("C_ad"), (tuple, list)):
        data["C_ad"] = f"[{data['C_ad'][0]}, {data['C_ad'][1]}]"
    if "Sim. duration" in data:
        data["Sim. duration"] = round(data["Sim. duration"], 2)
    missing = [f for f in fieldnames if f not in data]
    if missing:
        raise ValueError(f"results ledger of '{problem}': missing column(s) {missing}")
    csv_file_path = csv_file_path or LEDGER_FILES[problem]
    file_exists = os.path.isfile(csv_file_path)
    with open(csv_file_path, mode="a", newline="") as csv_file:
        writer = csv.DictWriter(csv_file, fieldnames=fieldnames)
        if not file_exists:
            writer.writeheader()
        writer.writerow(data)
    return csv_file_path
