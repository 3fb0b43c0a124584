"""per-kernel times of the benchmark window: python tools/kt.py [K] [ls_split]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from ddp_generator_amd import ilqg, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
split = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = 65536
x0, u0 = synth.car_batch(B, 500)
s = ilqg.BatchSolver("carparking", 0, batch=B, n_hor=500, params=ilqg.CAR_PARAMS, opts=dict(max_iter=K + 2, ls_split=split), groups=int(os.environ.get("NGROUPS", "0")))
s.init(x0, u0); s.iterate(2); s.sync(); s.init(x0, u0)
s.timing(False)
t0 = time.perf_counter(); s.iterate(K); s.sync(); dt = time.perf_counter() - t0
s.init(x0, u0); s.timing(True); s.iterate(K); s.sync()
kt = {k: round(v[1] / K, 3) for k, v in s.kernel_times().items() if v[0]}
print("%s K=%d split=%d: %.3f ms/iter %.1f it/s cost %.9f" % (os.environ.get("ILQG_LIBDIR", "lib"), K, split, 1e3 * dt / K, K / dt, s.scalar("cost").mean()), kt, flush=True)
s.close()
