"""where the clean and the poisoned run of test_failure_paths_are_per_trajectory_with_uniform_guards part"""
import sys, os, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
pkg = importlib.import_module("ddp-generator_amd")
ilqg = pkg.ilqg
import test_gpu_parity as T
fd = 0
B, N, iters = 70, 32, 2
x0, u0 = T.syn_inputs(B, N)
runs = []
for poison in (0, 1, 2, 3):
    u = u0.copy()
    if poison & 1:
        u[5, 10, 3] = np.nan
    s = ilqg.BatchSolver("synth16x8", fd, batch=B, n_hor=N, params=T.SYN_PARAMS_TIGHT, opts=dict(max_iter=iters + 1))
    s.init(x0, u)
    if poison & 2:
        x = s.x(); x[40, 7, 2] = np.inf; s.set_x(x)
    snap = []
    for it in range(iters):
        s.iterate(1)
        l, L = s.gains()
        snap.append(dict(l=l, L=L, x=s.x(), u=s.u(), cost=s.scalar("cost"), dV0=s.scalar("dV0"), lam=s.scalar("lambda"),
                         ac=s.scalar("alpha_cost"), ai=s.ints("alpha_idx"), bp=s.ints("bp_calls"), g=s.scalar("g_norm")))
    runs.append(snap)
    s.close()
ok = np.ones(B, dtype=bool); ok[[5, 40]] = False
for (na, a), (nb, b) in ((("clean", runs[0]), ("nan5", runs[1])), (("clean", runs[0]), ("inf40", runs[2])), (("clean", runs[0]), ("both", runs[3]))):
    for it in range(1):
        for k in a[it]:
            d = [i for i in np.where(ok)[0] if not np.array_equal(a[it][k][i], b[it][k][i], equal_nan=True)]
            if d:
                i = d[0]
                print(na, nb, "iteration", it, k, "differs for", d, "e.g. max abs diff", np.nanmax(np.abs(a[it][k][i] - b[it][k][i])))
