"""Config 5: what predicts the steps a trajectory's backward sweeps will walk in the NEXT iteration?  Dumps per iteration the
state before the pass (lambda, dlambda, previous steps / sweeps, accepted step index) and the steps after, for offline fits
and a simulation of the worker queue under different orders.  Needs the -DILQG_COUNT_STEPS build (see steps_hist.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N = 16384, 1000
x0, u0 = synth.synth16_batch(B, N)
s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=12))
s.init(x0, u0)
out = {}
for it in range(8):
    out["lam_%d" % it] = s.scalar("lambda").copy()
    out["dlam_%d" % it] = s.scalar("dlambda").copy()
    out["aidx_%d" % it] = s.ints("alpha_idx").copy()
    out["acc_%d" % it] = s.ints("accepted").copy()
    s.iterate(1)
    out["steps_%d" % it] = s.ints("bp_rc").copy()
    out["calls_%d" % it] = s.ints("bp_calls").copy()
    out["lam_after_%d" % it] = s.scalar("lambda").copy()
s.close()
np.savez_compressed(os.path.join("gpurun_out", "steps_predict.npz"), **out)
print("saved")
