"""results of a short synth16x8 FULL_DDP=0 solve under LDS poisoning variants (ILQG_LIBDIR selects the build)"""
import sys, os, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
pkg = importlib.import_module("ddp-generator_amd")
ilqg = pkg.ilqg
import test_gpu_parity as T
B, N, iters = 70, 32, 2
x0, u0 = T.syn_inputs(B, N)
s = ilqg.BatchSolver("synth16x8", 0, batch=B, n_hor=N, params=T.SYN_PARAMS_TIGHT, opts=dict(max_iter=iters + 1))
s.init(x0, u0)
s.iterate(1)
l, L = s.gains()
np.savez(sys.argv[1], cost=s.scalar("cost"), x=s.x(), u=s.u(), l=l, L=L, dV0=s.scalar("dV0"), g=s.scalar("g_norm"), lam=s.scalar("lambda"))
print("nan in l:", np.argwhere(np.isnan(l).any(axis=(1, 2))).ravel(), "nan in L:", np.argwhere(np.isnan(L).any(axis=(1, 2))).ravel())
