import sys, os, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
pkg = importlib.import_module("ddp-generator_amd")
ilqg = pkg.ilqg
import test_gpu_parity as T
fd = 0
B, N = 70, 32
x0, u0 = T.syn_inputs(B, N)
runs = []
for poison in (0, 1):
    u = u0.copy()
    if poison & 1:
        u[5, 10, 3] = np.nan
    s = ilqg.BatchSolver("synth16x8", fd, batch=B, n_hor=N, params=T.SYN_PARAMS_TIGHT, opts=dict(max_iter=3))
    s.init(x0, u)
    snap = dict(x0=s.x(), u0=s.u(), c0=s.scalar("cost"))
    if len(sys.argv) > 1 and sys.argv[1] == "stages":
        s.calc_derivs()
        rec, fin = s.derivs()
        snap.update(rec=rec, fin=fin)
        s.back_pass()
    else:
        s.iterate(1)
    l, L = s.gains()
    snap.update(l=l, L=L, dV0=s.scalar("dV0"), bp=s.ints("bp_calls"), lam=s.scalar("lambda"))
    runs.append(snap)
    s.close()
a, b = runs
for k in a:
    d = [i for i in range(B) if i != 5 and not np.array_equal(a[k][i], b[k][i], equal_nan=True)]
    print(k, "differs for", d)
    if d and a[k].ndim == 3:
        i = d[0]
        print("   steps", sorted(set(np.argwhere(a[k][i] != b[k][i])[:, 0].tolist())))
