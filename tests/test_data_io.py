"""Host I/O compatibility (CSV trajectories) -- CPU only."""
import importlib

import numpy as np


def test_csv_roundtrip_matches_reference_format(tmp_path):
    hp = importlib.import_module("fem-fct-pdeco_amd")
    V = hp.SquareMeshP1(0, 1, 4)
    n, Nt = V.nodes, 3
    rng = np.random.default_rng(0)
    traj = rng.random((Nt + 1) * n)
    f = tmp_path / "u.csv"
    hp.save_trajectory(f, traj)
    assert f.read_text().count("\n") == 0 and f.read_text().count(",") == traj.size - 1   # single line
    re_all, data_all = hp.import_data_final(f, n, V.vertex_to_dof, num_steps=Nt, time_dep=True)
    assert np.array_equal(data_all, traj)
    assert np.array_equal(re_all, hp.reorder_vector_from_dof(traj, Nt + 1, n, V.vertex_to_dof))
    re_T, data_T = hp.import_data_final(f, n, V.vertex_to_dof, num_steps=Nt)
    assert np.array_equal(data_T, traj[Nt * n:]) and re_T.shape == (5, 5)
    hp.extract_data(str(tmp_path), "u", T=0.2, dt=0.1, nodes=n)
    out = np.genfromtxt(tmp_path / "u_T0.2.csv", delimiter=",")
    assert np.allclose(out, traj[2 * n:3 * n], rtol=1e-15, atol=0)   # pandas' float parser (as in the reference): last digit


# ------------------------------------------------------------------- the reference's own files
# tests/golden/ref_data/ (make_golden.py --io-only): two data files AS THE REFERENCE WROTE THEM and what the real
# helpers.import_data_final / helpers.extract_data made of them in the build container.
import os

REF_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")


def test_import_data_final_reads_the_reference_target_file_like_the_reference():
    """data/solidbody_t0.25_u.csv, the final-time target of BASELINE config 2 (helpers.py:1874-1911)."""
    hp = importlib.import_module("fem-fct-pdeco_amd")
    z = np.load(os.path.join(REF_DATA, "io_ref.npz"))
    V = hp.SquareMeshP1(-1, 1, 80)
    re81, d81 = hp.import_data_final(os.path.join(REF_DATA, "solidbody_t0.25_u.csv"), V.nodes, V.vertex_to_dof)
    assert re81.shape == (81, 81) and np.array_equal(re81, z["re81"]) and np.array_equal(d81, z["d81"])
    # the committed .npz target used by bench.py / the parity tests holds the same numbers
    assert np.array_equal(d81, np.load(os.path.join(os.path.dirname(REF_DATA), "solidbody_t0.25_u.npz"))["u"])


def test_trajectory_file_of_the_reference_import_extract_and_rewrite(tmp_path):
    """First three time levels of the real-FEniCS chemotaxis trajectory file (bytes untouched)."""
    import shutil
    hp = importlib.import_module("fem-fct-pdeco_amd")
    z = np.load(os.path.join(REF_DATA, "io_ref.npz"))
    V = hp.SquareMeshP1(0, 1, 40)
    src = os.path.join(REF_DATA, "chtxs_m_3levels.csv")
    re_td, d_td = hp.import_data_final(src, V.nodes, V.vertex_to_dof, num_steps=2, time_dep=True)
    assert np.array_equal(re_td, z["re_td"]) and np.array_equal(d_td, z["d_td"])
    re_l1, d_l1 = hp.import_data_final(src, V.nodes, V.vertex_to_dof, num_steps=1)
    assert np.array_equal(re_l1, z["re_l1"]) and np.array_equal(d_l1, z["d_l1"])
    # extract_data writes the very bytes the reference's pandas + np.savetxt route wrote
    shutil.copyfile(src, tmp_path / "chtxs_m_3levels.csv")
    hp.extract_data(str(tmp_path), "chtxs_m_3levels", 0.002, 0.001, V.nodes, V.vertex_to_dof)
    assert (tmp_path / "chtxs_m_3levels_T0.002.csv").read_bytes() == \
        open(os.path.join(REF_DATA, "chtxs_m_3levels_T0.002.csv"), "rb").read()
    # save_trajectory re-creates the reference's file byte for byte from the parsed numbers
    hp.save_trajectory(tmp_path / "again.csv", d_td)
    assert (tmp_path / "again.csv").read_bytes() == open(src, "rb").read()


def test_results_ledger_and_result_files_follow_the_drivers(tmp_path):
    """Schnak_FCT_PDECO_refactored.py:271-299 (and the nonlinear / chemotaxis siblings): per-trajectory CSVs named
    <prefix>_<var>.csv and one appended ledger row per run, header only for a new file."""
    import csv
    hp = importlib.import_module("fem-fct-pdeco_amd")
    dio = importlib.import_module("fem-fct-pdeco_amd.data_io")
    rng = np.random.default_rng(1)
    u, c = rng.random(50), rng.random(50)
    paths = hp.save_results(str(tmp_path / "out"), "schnak", u=u, c=c)
    assert sorted(os.path.basename(p) for p in paths.values()) == ["AdvSchnak_c.csv", "AdvSchnak_u.csv"]
    assert np.array_equal(np.genfromtxt(paths["u"], delimiter=","), u)
    ledger = tmp_path / dio.LEDGER_FILES["schnak"]
    row = {"Sim. duration": 12.3456, "T": 0.5, "T_data": 0.5, "beta": 0.1, "tol": 1e-3, "GD its": 7,
           "Armijo its": [1, 2, 1], "C_ad": (0.0, 10.0), "Mean c. in L^2(Q)^2": 0.25, "Misfit norm u": 1e-3,
           "Misfit norm v": 2e-3, "J(c_true)": 0.05, "out_folder_name": "out"}
    hp.append_results_ledger("schnak", row, csv_file_path=str(ledger), timestamp="2025-01-01 00:00:00")
    hp.append_results_ledger("schnak", dict(row, **{"GD its": 8}), csv_file_path=str(ledger))
    rows = list(csv.reader(open(ledger, newline="")))
    assert rows[0] == ["timestamp", "Sim. duration", "T", "T_data", "beta", "tol", "GD its", "Armijo its", "C_ad",
                       "Mean c. in L^2(Q)^2", "Misfit norm u", "Misfit norm v", "J(c_true)", "out_folder_name"]
    assert len(rows) == 3 and rows[1][0] == "2025-01-01 00:00:00" and rows[1][1] == "12.35" and rows[1][8] == "[0.0, 10.0]"
    assert rows[1][6] == "7" and rows[2][6] == "8"
    import pytest
    with pytest.raises(ValueError):
        hp.append_results_ledger("schnak", {"T": 1.0}, csv_file_path=str(ledger))
    with pytest.raises(ValueError):
        hp.append_results_ledger("heat", row, csv_file_path=str(ledger))
    assert dio.LEDGER_FIELDS["chtxs"][-4:] == ["J(c_true)", "J_final_it", "J_diff", "out_folder_name"]
    assert "T_data" not in dio.LEDGER_FIELDS["chtxs"] and "Misfit norm" in dio.LEDGER_FIELDS["nonlinear"]
