import importlib, time, sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
hp = importlib.import_module("fem-fct-pdeco_amd"); systems = importlib.import_module("fem-fct-pdeco_amd.systems")
V = hp.SquareMeshP1(0.0, 1.0, 40); n = V.nodes; Nt, dt = 200, 5e-4
S = systems.PDESystems(V, order=hp.ORDER_VERTEX); ctx = S.ctx
tl = (Nt + 1) * n
rng = np.random.default_rng(0)
par, wind = systems._schnak_par(); Aw, AwT = S.convection(wind, "schnak")
u0s, v0s = hp.schnak_sys_IC(0, 1, 0.025, n, np.arange(n))
cpar = systems._chtxs_par()
eps, _, nwind = hp.get_nonlinear_eqns_params(); Awn, _ = S.convection(nwind, "nonlinear")
def timeit(fn, reps=3):
    fn(); fn(); ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize(); return (time.perf_counter() - t0) / reps
for B in (1, 4, 10, 20):
    init = np.zeros((B, tl)); init[:, :n] = u0s
    u = ctx.array(init.ravel()); init[:, :n] = v0s; v = ctx.array(init.ravel())
    c = ctx.array(0.1 + 0.01 * rng.random(B * n))
    ts = timeit(lambda: ctx.schnak_forward(Aw, c, u, v, Nt, dt, par, 1.0, batch=B))
    uc = ctx.array((1.5 + 0.1 * (0.5 - rng.random((B, 1))) * np.ones((B, tl))).ravel()); vc = ctx.array(np.zeros(B * tl) + 1.5)
    cc = ctx.array(20 * rng.random(B * n))
    tc = timeit(lambda: ctx.chtxs_forward(cc, uc, vc, Nt, dt, cpar, 0.1, batch=B))
    un = ctx.array(init.ravel())
    tn = timeit(lambda: ctx.nonlinear_forward(Awn, c, un, Nt, 1e-3, eps, batch=B))
    print(f"B={B:3d}  us/step: schnak {ts/Nt*1e6:7.1f}  chtxs {tc/Nt*1e6:7.1f}  nonlinear {tn/Nt*1e6:7.1f}", flush=True)
    for a in (u, v, c, uc, vc, cc, un): a.free()
