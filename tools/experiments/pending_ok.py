"""how many of the second stage's roll-outs of the n = 16 problem are not finite"""
import sys, os, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
pkg = importlib.import_module("ddp-generator_amd")
ilqg, synth = pkg.ilqg, pkg.synth
B, N = 16384, 1000
x0, u0 = synth.synth16_batch(B, N)
s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=6))
s.init(x0, u0)
for it in range(3):
    s.iterate(1)
    ok = s.ints("alpha_ok")[:, :8]
    idx = s.ints("alpha_idx")
    ac = s.scalar("alpha_cost")[:, :8]
    pend = idx != 1
    print("iteration", it, "pending", pend.sum(), "| roll-outs not finite per step size among them", (ok[pend] == 0).sum(axis=0),
          "| cost > 1e6:", (np.abs(ac[pend]) > 1e6).sum(axis=0), "| max |cost|", np.nanmax(np.abs(ac[pend]), axis=0).round(0))
