"""Config 5's backward kernel with every trajectory needing ONE sweep (a large initial lambda: no retries, 1 000 steps each, no
tail): what the two layouts of k_backward_quad do when nothing but throughput counts.
    ILQG_LIBDIR=... python tools/experiments/quad_equal_length.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N = int(os.environ.get("B", 16384)), 1000
x0, u0 = synth.synth16_batch(B, N)
for lam in (1e3, 1.0):
    s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=8, lambdaInit=lam, lambdaMax=1e12))
    s.init(x0, u0)
    for it in range(3):
        s.timing(True)
        s.iterate(1)
        s.sync()
        t = s.kernel_times()
        calls = s.ints("bp_calls")
        print("lambdaInit %g iteration %d: backward %.1f ms, derivs %.1f ms; sweeps mean %.2f max %d" % (lam, it + 1, t["k_backward"][1], t["k_derivs"][1], calls.mean(), calls.max()))
    s.close()
