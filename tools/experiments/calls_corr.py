"""Does the number of backward sweeps a trajectory needs in one iteration predict the next iteration's?  (config 5's problem)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N = 4096, 1000
x0, u0 = synth.synth16_batch(B, N)
s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=8))
s.init(x0, u0)
prev = None
for it in range(6):
    s.iterate(1)
    c = s.ints("bp_calls").copy()
    print("iteration", it + 1, "sweeps: mean %.2f" % c.mean(), "histogram", np.bincount(c)[:8], end="")
    if prev is not None:
        print("  corr with previous %.2f; mean now given previous == 1: %.2f, > 1: %.2f" % (np.corrcoef(prev, c)[0, 1], c[prev == 1].mean(), c[prev > 1].mean()))
    else:
        print()
    prev = c
s.close()
