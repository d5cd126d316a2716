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
    assert np.allclose(out, traj[2 * n:3 * n], rtol=0, atol=0)
