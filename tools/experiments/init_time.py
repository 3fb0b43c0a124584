import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N = 16384, 1000
x0, u0 = synth.synth16_batch(B, N)
for opts in (dict(), dict(ls_split=0)):
    s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=4, **opts))
    s.init(x0, u0); s.sync()
    t0 = time.perf_counter(); s.init(x0, u0); s.sync(); t1 = time.perf_counter()
    s.iterate(2); s.sync(); t2 = time.perf_counter()
    print(opts, "init %.1f ms, 2 iterations %.1f ms, cost %.12g" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), s.scalar("cost").mean()))
    s.close()
