"""Which step size each trajectory accepts, per iteration (decides how the line search should be staged).
python tools/alpha_hist.py [carparking|synth16x8]"""
import sys, os, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ddp-generator_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "carparking"
if name == "carparking":
    B, N, fd = 16384, 500, 0
    x0, u0 = pkg.synth.car_batch(B, N)
    params = pkg.ilqg.CAR_PARAMS
else:
    B, N, fd = 1024, 1000, 1
    x0, u0 = pkg.synth.synth16_batch(B, N)
    params = pkg.synth.SYNTH16_PARAMS
s = pkg.ilqg.BatchSolver(name, fd, B, N, params=params)
s.init(x0, u0)
tot = np.zeros(10, dtype=np.int64)
for it in range(20):
    s.iterate(1)
    s.sync()
    acc = s.ints("accepted")
    idx = s.ints("alpha_idx")
    h = np.bincount(np.where(acc > 0, idx, 9), minlength=10)
    tot += h
    print(it, "accepted at alpha index 0..7 | 8 unused | 9 = none:", (h / B).round(3))
print("all:", (tot / tot.sum()).round(4))
