"""experiment: two chunks of the batch on separate streams, the second trailing by one phase"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from ddp_generator_amd import ilqg, synth
K = 10

def run(B, nch, split, stagger):
    x0, u0 = synth.car_batch(B)
    per = B // nch
    ss = [ilqg.BatchSolver("carparking", 0, batch=per, n_hor=500, params=ilqg.CAR_PARAMS,
                           opts=dict(max_iter=K + 2, ls_split=split)) for _ in range(nch)]
    def init():
        for i, s in enumerate(ss):
            s.init(x0[i * per:(i + 1) * per], u0[i * per:(i + 1) * per])
        for s in ss: s.sync()
    init()
    for s in ss: s.iterate(1)
    for s in ss: s.sync()
    init()
    t0 = time.perf_counter()
    if stagger:
        for j, s in enumerate(ss):
            for _ in range(j):
                s.back_pass(fused=True)   # delays chunk j by j backward passes (same result)
    for it in range(K):
        for s in ss:
            s.iterate(1)
    for s in ss: s.sync()
    dt = time.perf_counter() - t0
    c = np.mean([s.scalar("cost").mean() for s in ss])
    print("B %6d chunks %d split %d stagger %d: %.2f ms/iter  %.1f it/s  cost mean %.6f" % (B, nch, split, stagger, 1e3 * dt / K, K / dt, c), flush=True)
    for s in ss: s.close()

import sys
for split in (5, 3, 2):
    run(65536, 1, split, 0)
    run(65536, 2, split, 0)
    run(65536, 2, split, 1)
    run(65536, 4, split, 0)
    run(65536, 4, split, 1)
