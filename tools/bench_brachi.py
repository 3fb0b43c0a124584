"""Throughput of the multiplier path: B Brachistochrone trajectories (hli + hfe constraints, n = 1, m = 1, N = 500)
in lock step, full solves.  Prints one JSON line.  Usage: python tools/bench_brachi.py [B]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from ddp_generator_amd import ilqg  # noqa: E402
from oracle.harness import brachi_hli_case  # noqa: E402  (parameters of the reference's demo only)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = 500
params, opts, _, _ = brachi_hli_case(n)
rng = np.random.default_rng(1)
x0 = -10.0 ** rng.uniform(-16, -2, (B, 1))
u0 = -np.ones((B, n, 1)) * rng.uniform(0.5, 1.5, (B, 1, 1))
s = ilqg.BatchSolver("brachi_hli", 0, batch=B, n_hor=n, params=params, opts=opts)
s.init(x0, u0)
s.sync()
s.timing(True)
t0 = time.perf_counter()
s.solve()
s.sync()
dt = time.perf_counter() - t0
it = s.ints("iterations")
ok = s.success()
yN = s.x()[:, -1, 0]
out = dict(workload="Brachistochrone hli+hfe, batch %d, N=%d, full solves (max_iter %d)" % (B, n, opts["max_iter"]),
           seconds=dt, trajectories_per_s=B / dt, iterations_mean=float(it.mean()), iterations_max=int(it.max()),
           lockstep_iterations_per_s=float(it.max()) / dt, converged=float(ok.mean()),
           y_final_mean=float(yN.mean()), kernels_ms={k: round(v[1], 3) for k, v in s.kernel_times().items() if v[0]})
print(json.dumps(out))
