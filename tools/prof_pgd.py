import importlib, time, sys, os, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
hp = importlib.import_module("fem-fct-pdeco_amd"); pdeco = importlib.import_module("fem-fct-pdeco_amd.pdeco")
V = hp.SquareMeshP1(0.0, 1.0, 40); n = V.nodes; Nt, dt = 200, 5e-4; tl=(Nt+1)*n
ic = hp.chtxs_sys_IC(0, 1, 0.025, n, V.vertex_to_dof)
z = lambda x0: np.concatenate([x0, np.zeros(Nt * n)])
full = hp.solve_chtxs_system(np.full(tl, 10.0), z(ic[0]), z(ic[1]), V, n, Nt, dt, None)
tg = tuple(np.array(f) for f in full)
with pdeco.SystemPDECO("chtxs", V, Nt, dt, max_iter_GD=3, tol=0.0) as P:
    P.run(ic, tg, speculative=True)
with pdeco.SystemPDECO("chtxs", V, Nt, dt, max_iter_GD=3, tol=0.0) as P:
    pr = cProfile.Profile(); pr.enable()
    t0=time.perf_counter(); r = P.run(ic, tg, speculative=True); el=time.perf_counter()-t0
    pr.disable()
print("total", el, "its", r["it"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
