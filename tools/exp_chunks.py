"""experiment: does running chunks of the batch on separate streams raise throughput?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from ddp_generator_amd import ilqg, synth
B, K = 65536, 10
x0, u0 = synth.car_batch(B)
for nch in (1, 2, 4, 8, 16):
    for split in (3, 4):
        per = B // nch
        ss = [ilqg.BatchSolver("carparking", 0, batch=per, n_hor=500, params=ilqg.CAR_PARAMS,
                               opts=dict(max_iter=K + 2, ls_split=split)) for _ in range(nch)]
        for i, s in enumerate(ss):
            s.init(x0[i * per:(i + 1) * per], u0[i * per:(i + 1) * per])
        for s in ss: s.iterate(1)
        for s in ss: s.sync()
        for i, s in enumerate(ss):
            s.init(x0[i * per:(i + 1) * per], u0[i * per:(i + 1) * per])
        for s in ss: s.sync()
        t0 = time.perf_counter()
        for it in range(K):
            for s in ss: s.iterate(1)
        for s in ss: s.sync()
        dt = time.perf_counter() - t0
        print("chunks %2d split %d: %.2f ms/iter  %.1f it/s  cost mean %.6f" % (nch, split, 1e3 * dt / K, K / dt, np.mean([s.scalar("cost").mean() for s in ss])), flush=True)
        for s in ss: s.close()
