import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
for fd, B, N in ((0, 1024, 1000), (1, 1024, 1000), (1, 256, 1000)):
    x0, u0 = synth.synth16_batch(B, N)
    s = ilqg.BatchSolver("synth16x8", fd, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=10))
    s.init(x0, u0)
    s.timing(True)
    calls = []
    for it in range(3):
        s.iterate(1)
        calls.append(s.ints("bp_calls").copy())
    s.sync()
    t = s.kernel_times()
    print("fd", fd, "B", B, "N", N, {k: (v[0], round(v[1] / 3, 2)) for k, v in t.items() if v[0]})
    print("   bp_calls per iteration: max", [int(c.max()) for c in calls], "mean", [round(float(c.mean()), 2) for c in calls],
          "alpha idx", np.bincount(s.ints("alpha_idx"), minlength=10)[:10], "status", np.bincount(s.ints("status"), minlength=8))
    s.close()
