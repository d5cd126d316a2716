#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run ONLY in the build container (needs /root/reference, which never travels to
the GPU box):

    python tests/golden/make_golden.py

It imports the reference's ``helpers.py`` unmodified (with an empty stand-in
module named ``dolfin``; the FCT core never touches dolfin), feeds it seeded
inputs, and stores inputs + the reference's outputs as small ``.npz`` files.
It also re-saves two of the reference's own data files (real-FEniCS outputs /
targets) as ``.npz``.  Only data is written here -- no reference source text.

Files written
  fct_cases.npz            FCT_alg_ref input/output pairs (helpers.py:1715-1872)
  fct_old_sign.npz         old_helpers.FCT_alg vs FCT_alg_ref (old_helpers.py:115-203)
  kernels.npz              ChebSI, artificial_diffusion_mat, row_lump, sparse_nonzero,
                           L2 norms, cost_functional (helpers.py:143-441)
  solidbody_traj_N21.npz   20 forward + 20 adjoint reference FCT steps, drift control
  chtxs_fenics_traj.npz    Chtxs_data_dx0.025_dt0.001/chtxs_{m,f}_t0.01.csv (real FEniCS)
  solidbody_t0.25_u.npz    data/solidbody_t0.25_u.csv (PDECO target, an input)
  ref_data/                (``--io-only`` regenerates just these) two of the reference's DATA files in the
                           reference's own on-disk format -- data/solidbody_t0.25_u.csv whole, the first three
                           time levels of Chtxs_data_dx0.025_dt0.001/chtxs_m_t0.01.csv (a byte prefix cut at a
                           comma) -- plus what the REAL import_data_final / extract_data (helpers.py:1874-1956)
                           return / write for them: io_ref.npz, chtxs_m_3levels_T0.002.csv
"""
import io
import os
import sys
import types
import contextlib

import numpy as np
from scipy.sparse import csr_matrix, lil_matrix, diags

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def import_reference():
    stub = types.ModuleType("dolfin")
    for a in ("dx", "dot", "grad", "assemble", "exp"):
        setattr(stub, a, None)
    sys.modules["dolfin"] = stub
    sys.path.insert(0, REF)
    import helpers as hp  # noqa
    return hp


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def main():
    hp = import_reference()
    from oracle.mesh import SquareMesh
    from oracle.assembly import P1Assembler, row_lump_diag
    from oracle.traj import rotation_wind, schnak_wind

    rng = np.random.default_rng(20261004)

    def setup(a1, a2, n_cells):
        mesh = SquareMesh(a1, a2, n_cells)
        asm = P1Assembler(mesh)
        M = asm.mass()
        Ad = asm.stiffness()
        return mesh, asm, M, Ad

    def csr_pack(prefix, A, out):
        A = csr_matrix(A)
        A.sort_indices()
        out[prefix + "_data"] = A.data
        out[prefix + "_indices"] = A.indices.astype(np.int32)
        out[prefix + "_indptr"] = A.indptr.astype(np.int32)

    # ------------------------------------------------------------------ FCT
    cases = {}
    names = []

    def add_case(name, a1, a2, n_cells, build):
        mesh, asm, M, Ad = setup(a1, a2, n_cells)
        n = mesh.nodes
        A, rhs, u_n, dt, nfm = build(mesh, asm, M, Ad)
        Ml = hp.row_lump(lil_matrix(M), n)
        nbrs = mesh.dof_neighbors()
        u_np1, log = quiet(hp.FCT_alg_ref, csr_matrix(A), rhs, u_n, dt, n, lil_matrix(M), Ml, nbrs,
                           non_flux_mat=None if nfm is None else csr_matrix(nfm))
        d = {}
        d["geom"] = np.array([a1, a2, n_cells], dtype=np.float64)
        csr_pack("A", A, d)
        csr_pack("M", M, d)
        d["ml"] = Ml.diagonal()
        d["rhs"] = rhs
        d["u_n"] = u_n
        d["dt"] = np.float64(dt)
        d["has_nfm"] = np.int32(nfm is not None)
        if nfm is not None:
            csr_pack("N", nfm, d)
        d["u_np1"] = np.asarray(u_np1)
        d["mmatrix_failed"] = np.int32("3: False" in log)
        for k, v in d.items():
            cases[f"{name}/{k}"] = v
        names.append(name)
        print(f"  case {name}: n={n}, M-matrix check failed={bool(d['mmatrix_failed'])}")

    def bump(mesh, noise=0.05):
        v2d = mesh.vertex_to_dof
        u = np.exp(-8 * ((mesh.x - 0.2 * (mesh.a1 + mesh.a2)) ** 2 + (mesh.y - 0.5 * (mesh.a1 + mesh.a2)) ** 2))
        u = u + noise * rng.random(mesh.nodes)
        out = np.empty(mesh.nodes)
        out[v2d] = u
        return out

    om = np.pi / 40

    def rotation(mesh, asm, M, Ad):
        # FCT_alg(A_u) == FCT_alg_ref(-A_u), A_u = Arot (finaltime.py:191-193)
        return -asm.convection(rotation_wind(om)), np.zeros(mesh.nodes), bump(mesh), 1e-3, None

    def rotation_drift22(mesh, asm, M, Ad):
        w = lambda x, y: (-(1 / om) * y + 2.0, (1 / om) * x + 2.0)
        return -asm.convection(w), np.zeros(mesh.nodes), bump(mesh), 1e-3, None

    def rotation_drift22_bigdt(mesh, asm, M, Ad):
        # dt large enough that the diagonal-dominance check (helpers.py:1796-1809) fails
        w = lambda x, y: (-(1 / om) * y + 2.0, (1 / om) * x + 2.0)
        return -asm.convection(w), np.zeros(mesh.nodes), bump(mesh), 4e-3, None

    def drift_control(mesh, asm, M, Ad):
        c = 5 * rng.random(mesh.nodes)
        A_u = asm.convection(rotation_wind(om)) + asm.drift1(c) + asm.drift2(c)
        return -A_u, np.zeros(mesh.nodes), bump(mesh), 1e-3, None

    def schnak_like(mesh, asm, M, Ad):
        A = asm.convection(schnak_wind)
        Mat = 0.01 * Ad - 100 * A
        rhs = asm.load(lambda at: 230.82 * (0.1 + at(bump(mesh)) ** 2))
        return Mat, rhs, 1.0 + bump(mesh), 1e-3, 230.82 * M

    def nonlinear_like(mesh, asm, M, Ad):
        A = asm.convection(lambda x, y: (2 * (y - .5) * x * (1 - x), -2 * (x - .5) * y * (1 - y)))
        u = bump(mesh)
        Mu2 = asm.weighted_mass(lambda at: at(u) ** 2)
        return -(A - 1e-4 * Ad), asm.load(lambda at: at(0.3 * bump(mesh))), u, 5e-3, -M + Mu2 / 3

    def chtxs_like(mesh, asm, M, Ad):
        u = 1.5 + 0.1 * (0.5 - rng.random(mesh.nodes))
        v = 1.5 + 0.1 * (0.5 - rng.random(mesh.nodes))
        Aa = asm.chtxs_forward_Aa(u, v, 0.5)
        return 0.05 * Ad - 0.25 * Aa, np.zeros(mesh.nodes), u, 1e-3, None

    def adjoint_rhs(mesh, asm, M, Ad):
        c = 5 * rng.random(mesh.nodes)
        A_p = -asm.convection(rotation_wind(om)) - asm.drift1(c) - asm.drift2(c)
        rhs = asm.load(lambda at: at(bump(mesh)) - at(bump(mesh)))
        return -A_p, rhs, bump(mesh) - 0.5, 1e-3, None

    print("FCT_alg_ref cases")
    add_case("rot_N5", -1, 1, 4, rotation)
    add_case("rot_N11", -1, 1, 10, rotation)
    add_case("rot_N41", -1, 1, 40, rotation)
    add_case("rotdrift22_N11", -1, 1, 10, rotation_drift22)
    add_case("rotdrift22_N41", -1, 1, 40, rotation_drift22)
    add_case("rotdrift22_bigdt_N41", -1, 1, 40, rotation_drift22_bigdt)
    add_case("driftctl_N11", -1, 1, 10, drift_control)
    add_case("driftctl_N41", -1, 1, 40, drift_control)
    add_case("schnak_N11", 0, 1, 10, schnak_like)
    add_case("schnak_N41", 0, 1, 40, schnak_like)
    add_case("nonlinear_N11", 0, 1, 10, nonlinear_like)
    add_case("chtxs_N11", 0, 1, 10, chtxs_like)
    add_case("chtxs_N41", 0, 1, 40, chtxs_like)
    add_case("adjrhs_N11", -1, 1, 10, adjoint_rhs)
    cases["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "fct_cases.npz"), **cases)

    # ------------------------------------------------- old sign convention
    src = open(os.path.join(REF, "old_helpers.py")).read()
    ns = dict(vars(hp))
    exec(compile(src, "old_helpers.py", "exec"), ns)
    mesh, asm, M, Ad = setup(0, 1, 10)
    n = mesh.nodes
    A = asm.convection(schnak_wind) * 50 - 0.01 * Ad
    S = 0.3 * M
    u_n = 1 + bump(mesh)
    rhs = asm.load(lambda at: at(bump(mesh)))
    Ml = hp.row_lump(lil_matrix(M), n)
    nb = mesh.dof_neighbors()
    u_old, _ = quiet(ns["FCT_alg"], csr_matrix(A), rhs, u_n, 2e-3, n, lil_matrix(M), Ml, nb, source_mat=csr_matrix(S))
    u_new, _ = quiet(hp.FCT_alg_ref, csr_matrix(-A), rhs, u_n, 2e-3, n, lil_matrix(M), Ml, nb, non_flux_mat=csr_matrix(S))
    d = dict(geom=np.array([0, 1, 10.0]), rhs=rhs, u_n=u_n, dt=np.float64(2e-3), u_old=u_old, u_new=u_new)
    csr_pack("A", A, d)
    csr_pack("S", S, d)
    np.savez_compressed(os.path.join(HERE, "fct_old_sign.npz"), **d)
    print("old-sign FCT_alg vs FCT_alg_ref max diff:", np.max(np.abs(u_old - u_new)))

    # ---------------------------------------------------- small kernels
    mesh, asm, M, Ad = setup(-1, 1, 10)
    n = mesh.nodes
    k = {}
    b = rng.standard_normal(n)
    k["geom"] = np.array([-1, 1, 10.0])
    k["cheb_b"] = b
    k["cheb_y"] = hp.ChebSI(b, M, M.diagonal(), 20, 0.5, 2)
    K = asm.convection(rotation_wind(om)) + asm.drift1(5 * rng.random(n)) + 0.3 * Ad
    csr_pack("K", K, k)
    Dref = csr_matrix(hp.artificial_diffusion_mat(lil_matrix(K)))
    csr_pack("D", Dref, k)
    k["ml"] = hp.row_lump(lil_matrix(M), n).diagonal()
    nz = hp.sparse_nonzero(csr_matrix(K))
    k["K_nonzero"] = nz
    Nt = 6
    phi = rng.standard_normal((Nt + 1) * n)
    tgt = rng.standard_normal((Nt + 1) * n)
    ctl = rng.random((Nt + 1) * n)
    phi2 = rng.standard_normal((Nt + 1) * n)
    tgt2 = rng.standard_normal((Nt + 1) * n)
    k["phi"], k["tgt"], k["ctl"], k["phi2"], k["tgt2"] = phi, tgt, ctl, phi2, tgt2
    k["Nt"] = np.int32(Nt)
    k["dt"] = np.float64(0.01)
    k["beta"] = np.float64(0.1)
    k["L2Q"] = hp.L2_norm_sq_Q(phi, Nt, 0.01, M)
    k["L2Omega"] = hp.L2_norm_sq_Omega(phi[:n], M)
    k["J_alltime_1"] = quiet(hp.cost_functional, phi, tgt, ctl, Nt, 0.01, M, 0.1, "alltime")[0]
    k["J_alltime_2"] = quiet(hp.cost_functional, phi, tgt, ctl, Nt, 0.01, M, 0.1, "alltime", var2=phi2, var2_target=tgt2)[0]
    k["J_finaltime_1"] = quiet(hp.cost_functional, phi, tgt[:n], ctl, Nt, 0.01, M, 0.1, "finaltime")[0]
    k["J_finaltime_2"] = quiet(hp.cost_functional, phi, tgt[:n], ctl, Nt, 0.01, M, 0.1, "finaltime", var2=phi2, var2_target=tgt2[:n])[0]
    v2d = mesh.vertex_to_dof
    vv = rng.standard_normal(2 * n)
    k["reorder_in"] = vv
    k["reorder_to"] = hp.reorder_vector_to_dof(vv, 2, n, v2d)
    k["reorder_from"] = hp.reorder_vector_from_dof(vv, 2, n, v2d)
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **k)

    # ----------------------- short drift-control trajectory through the real FCT
    mesh, asm, M, Ad = setup(-1, 1, 20)
    n = mesh.nodes
    Nt, dt = 20, 2e-3
    Ml = hp.row_lump(lil_matrix(M), n)
    Mlil = lil_matrix(M)
    nb = mesh.dof_neighbors()
    Arot = asm.convection(rotation_wind(om))
    ck = 2.0 * rng.random((Nt + 1) * n)
    uk = np.zeros((Nt + 1) * n)
    X, Y = mesh.x, mesh.y
    u0 = ((np.sqrt(X ** 2 + (Y - 1 / 3) ** 2) < 1 / 3) & ((np.abs(X) > 0.05) | (Y > 0.5))).astype(float)
    uk[:n][v2d_ := mesh.vertex_to_dof] = u0
    for i in range(1, Nt + 1):
        c = ck[i * n:(i + 1) * n]
        A_u = Arot + asm.drift1(c) + asm.drift2(c)
        uk[i * n:(i + 1) * n], _ = quiet(hp.FCT_alg_ref, csr_matrix(-A_u), np.zeros(n), uk[(i - 1) * n:i * n],
                                          dt, n, Mlil, Ml, nb)
    uhat = uk[Nt * n:] * 0.9 + 0.05
    pk = np.zeros((Nt + 1) * n)
    pk[Nt * n:] = uhat - uk[Nt * n:]
    for i in reversed(range(Nt)):
        c = ck[i * n:(i + 1) * n]
        A_p = -Arot - asm.drift1(c) - asm.drift2(c)
        pk[i * n:(i + 1) * n], _ = quiet(hp.FCT_alg_ref, csr_matrix(-A_p), np.zeros(n), pk[(i + 1) * n:(i + 2) * n],
                                          dt, n, Mlil, Ml, nb)
    np.savez_compressed(os.path.join(HERE, "solidbody_traj_N21.npz"), geom=np.array([-1, 1, 20.0]),
                        Nt=np.int32(Nt), dt=np.float64(dt), om=np.float64(om), ck=ck, uk=uk, uhat=uhat, pk=pk)
    print("solid-body trajectory: mass drift", abs(Ml.diagonal() @ uk[Nt * n:] - Ml.diagonal() @ uk[:n]))

    # --------------------------------------------- reference data files
    m = np.genfromtxt(os.path.join(REF, "Chtxs_data_dx0.025_dt0.001/chtxs_m_t0.01.csv"), delimiter=",")
    f = np.genfromtxt(os.path.join(REF, "Chtxs_data_dx0.025_dt0.001/chtxs_f_t0.01.csv"), delimiter=",")
    np.savez_compressed(os.path.join(HERE, "chtxs_fenics_traj.npz"), m=m, f=f)
    t = np.genfromtxt(os.path.join(REF, "data/solidbody_t0.25_u.csv"), delimiter=",")
    np.savez_compressed(os.path.join(HERE, "solidbody_t0.25_u.npz"), u=t)
    print("done")


def data_io_fixtures(hp=None):
    """Reference-written data files + the reference's own readers' outputs (host I/O compatibility)."""
    import shutil
    hp = hp or import_reference()
    from oracle.mesh import SquareMesh
    out = os.path.join(HERE, "ref_data")
    os.makedirs(out, exist_ok=True)
    # (1) final-time target of config C2, exactly as the reference ships it
    src = os.path.join(REF, "data/solidbody_t0.25_u.csv")
    shutil.copyfile(src, os.path.join(out, "solidbody_t0.25_u.csv"))
    os.chmod(os.path.join(out, "solidbody_t0.25_u.csv"), 0o644)
    v2d81 = SquareMesh(-1, 1, 80).vertex_to_dof
    re81, d81 = hp.import_data_final(src, 6561, v2d81)
    # (2) a trajectory file: first 3 levels (3 * 1681 values) of the real-FEniCS chemotaxis trajectory, bytes untouched
    raw = open(os.path.join(REF, "Chtxs_data_dx0.025_dt0.001/chtxs_m_t0.01.csv"), "rb").read()
    pos = -1
    for _ in range(3 * 1681):
        pos = raw.index(b",", pos + 1)
    open(os.path.join(out, "chtxs_m_3levels.csv"), "wb").write(raw[:pos])
    v2d41 = SquareMesh(0, 1, 40).vertex_to_dof
    traj = os.path.join(out, "chtxs_m_3levels.csv")
    re_td, d_td = hp.import_data_final(traj, 1681, v2d41, num_steps=2, time_dep=True)
    re_l1, d_l1 = hp.import_data_final(traj, 1681, v2d41, num_steps=1)
    # (3) the reference's extract_data on it (writes <name>_T<T>.csv next to the input)
    quiet(hp.extract_data, out, "chtxs_m_3levels", 0.002, 0.001, 1681, v2d41)
    np.savez_compressed(os.path.join(out, "io_ref.npz"), re81=re81, d81=d81, re_td=re_td, d_td=d_td, re_l1=re_l1, d_l1=d_l1)
    print("ref_data written:", sorted(os.listdir(out)))


if __name__ == "__main__":
    if "--io-only" in sys.argv:
        data_io_fixtures()
    else:
        main()
        data_io_fixtures()
