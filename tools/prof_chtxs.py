import importlib, sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
hp = importlib.import_module("fem-fct-pdeco_amd"); systems = importlib.import_module("fem-fct-pdeco_amd.systems")
V = hp.SquareMeshP1(0.0, 1.0, 40); n = V.nodes; Nt, dt = 200, 5e-4
S = systems.PDESystems(V, order=hp.ORDER_VERTEX); ctx = S.ctx
tl = (Nt + 1) * n; rng = np.random.default_rng(0)
cpar = systems._chtxs_par()
u0 = 1.5 + 0.1 * (0.5 - rng.random(n))
z = lambda x0: np.concatenate([x0, np.zeros(Nt * n)])
u, v = ctx.array(z(u0)), ctx.array(z(u0)); cc = ctx.array(20 * rng.random(n))
for _ in range(3): ctx.chtxs_forward(cc, u, v, Nt, dt, cpar, 0.1)
