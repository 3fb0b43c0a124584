"""Eight independent chains instead of four: two contexts of 32 768 CarParking trajectories (four stream groups each) iterated
alternately from one host thread, against one context of 65 536 — with the process's default four hardware queues and with
GPU_MAX_HW_QUEUES=8 (set in the environment before the first HIP call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N, K, W = 65536, 500, 20, 2
parts = int(os.environ.get("PARTS", 2))
x0, u0 = synth.car_batch(B, N)
per = B // parts
ss = []
for p in range(parts):
    s = ilqg.BatchSolver("carparking", 0, batch=per, n_hor=N, params=ilqg.CAR_PARAMS, opts=dict(max_iter=K + W + 2))
    s.init(x0[p * per:(p + 1) * per], u0[p * per:(p + 1) * per])
    ss.append(s)
for _ in range(W):
    for s in ss: s.iterate(1)
for s in ss: s.sync()
t0 = time.perf_counter()
for _ in range(K):
    for s in ss: s.iterate(1)
for s in ss: s.sync()
dt = time.perf_counter() - t0
print("%d context(s) x %d groups, GPU_MAX_HW_QUEUES=%s: %.1f it/s (%.2f ms); cost mean %.6f" % (parts, ss[0].groups(), os.environ.get("GPU_MAX_HW_QUEUES"), K / dt, 1e3 * dt / K,
      np.mean([s.scalar("cost").mean() for s in ss])))
for s in ss: s.close()
