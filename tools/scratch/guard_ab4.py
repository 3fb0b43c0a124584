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
poison = int(sys.argv[1])
u = u0.copy()
if poison == 1:
    u[5, 10, 3] = np.nan
if poison == 2:   # trajectory 5 merely different, not failing
    u[5, 10, 3] += 0.5
s = ilqg.BatchSolver("synth16x8", fd, batch=B, n_hor=N, params=T.SYN_PARAMS_TIGHT, opts=dict(max_iter=3))
s.init(x0, u)
s.iterate(1)
l, L = s.gains()
np.savez(sys.argv[2], l=l, L=L, bp=s.ints("bp_calls"), st=s.ints("status"), rc=s.ints("bp_rc"))
