"""tools/pmc_traffic.py reduces two rocprofv3 --pmc CSVs to bytes per launch and class (CPU only, synthetic input)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for name, grid, val in rows:
            w.writerow(dict(Kernel_Name=name, Grid_Size=grid, Counter_Name=counter, Counter_Value=val))


def test_pmc_traffic_classes_and_split_chebyshev_launch(tmp_path):
    n = 4198401
    ns = "(anonymous namespace)::"
    fetch = [(ns + "k_strip4_jacobi_walk(int)", 256 * 1024, 100.0), (ns + "k_strip4_jacobi_walk(int)", 256 * 1024, 300.0),
             (ns + "k_strip4_cheb_mass_int(int)", 2025 * 1024, 50.0),
             ("void " + ns + "k_strip4_cheb_mass<1>(int)", 184 * 1024, 5.0),      # the boundary ring of the same launch
             ("void " + ns + "k_strip4_cheb_mass<0>(int)", 4 * 1024, 999.0),       # a small mesh: ignored
             (ns + "k_strip4_cheb_mass_walk(int)", 256 * 1024, 7.0),              # superseded by the interior kernel
             ("void " + ns + "k_build_low_sb<256>(int)", 2048 * 256, 40.0),
             ("void " + ns + "k_jacobi<7, 256, 1>(int)", 2048 * 256, 10.0)]
    write = [(k, g, v / 10.0) for k, g, v in fetch]
    fp, wp = tmp_path / "fetch.csv", tmp_path / "write.csv"
    _write(fp, "FETCH_SIZE", fetch)
    _write(wp, "WRITE_SIZE", write)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(fp), str(wp), str(n),
                          str(tmp_path / "t")], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 0, out.stderr[-2000:]
    t = json.load(open(tmp_path / "traffic.json"))[f"n{n}"]
    kib = 1024.0
    b = t["bytes_per_launch"]
    assert abs(b["jacobi"] - (2 * 200.0 + 20.0) * kib) < 1e-6                      # mean over the two launches
    assert abs(b["cheb"] - ((2 * 50.0 + 5.0) + (2 * 5.0 + 0.5)) * kib) < 1e-6      # interior kernel + its ring
    assert abs(b["build_low"] - (2 * 40.0 + 4.0) * kib) < 1e-6
    assert abs(t["one_sweep_bytes_per_launch"]["jacobi"] - (2 * 10.0 + 1.0) * kib) < 1e-6
    assert len(t["source_sha16"]) == 16


def test_bench_bytes_model_of_the_bandwidth_regime():
    """bench.launch_bytes_per_row: what a launch must move per matrix row (the numerator of roofline.frac) under the
    traffic savers of the bandwidth regime -- operator derived in the kernels, its rotation part from the node positions,
    D once per edge, the zeros of L neither stored nor loaded."""
    sys.path.insert(0, ROOT)
    import bench
    base = bench.launch_bytes_per_row(fused=True, geom_mass=True)
    assert base["jacobi"] == 80 and base["build_low"] == 200 and base["dudt_rhs"] == 96 and base["cheb"] == 32
    inl = bench.launch_bytes_per_row(fused=True, geom_mass=True, inline_ops=True, half_d=True, l_nonzero=0.5)
    rot = bench.launch_bytes_per_row(fused=True, geom_mass=True, inline_ops=True, half_d=True, l_nonzero=0.5, rot_geom=True)
    assert inl["build_low"] - rot["build_low"] == 56 and inl["dudt_rhs"] - rot["dudt_rhs"] == 56      # Arot: 7 doubles per row
    assert rot["dudt_rhs"] == 48 and abs(rot["build_low"] - 97) < 1e-9
    assert abs(inl["jacobi"] - (8 + 24 + 24 + 1)) < 1e-9 and inl["flux"] == 56
    # without the inline operator the rotation flag has nothing to act on
    assert bench.launch_bytes_per_row(fused=True, geom_mass=True, rot_geom=True) == base
