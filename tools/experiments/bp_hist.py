"""sweeps per backward pass (lambda retries) of the headline batch, per iteration"""
import sys, os, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
pkg = importlib.import_module("ddp-generator_amd")
ilqg, synth = pkg.ilqg, pkg.synth
B, N = 16384, 500
x0, u0 = synth.car_batch(B, N)
s = ilqg.BatchSolver("carparking", 0, batch=B, n_hor=N, params=ilqg.CAR_PARAMS, opts=dict(max_iter=25))
s.init(x0, u0)
for it in range(20):
    s.iterate(1)
    c = s.ints("bp_calls")
    h = np.bincount(np.minimum(c, 8), minlength=9)
    per_wave = c.reshape(-1, 64).max(axis=1)
    print(it, "sweeps per trajectory 0..8+:", h, "| mean", c.mean().round(3), "| per wavefront (max of 64): mean", per_wave.mean().round(2), "max", per_wave.max())
