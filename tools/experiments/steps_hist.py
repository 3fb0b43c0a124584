"""Config 5's problem: how many steps do the backward sweeps of a trajectory walk per iteration (lambda retries abandon a sweep
where the box QP fails, back_pass.c:168-171)?  The longest trajectory is the backward kernel's critical path: its steps are a
serial chain whatever the chip does with the rest.  Needs a -DILQG_COUNT_STEPS build (bp_rc then reports the steps walked):
    tools/variant.sh steps "-DILQG_COUNT_STEPS=1" synth16x8 1;  ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_steps python tools/experiments/steps_hist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N = int(os.environ.get("B", 16384)), 1000
x0, u0 = synth.synth16_batch(B, N)
s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=8))
s.init(x0, u0)
prev = None
for it in range(5):
    s.timing(True)
    s.iterate(1)
    s.sync()
    ms = s.kernel_times().get("k_backward", (0, 0.0))[1]
    st, calls = s.ints("bp_rc").copy().astype(np.int64), s.ints("bp_calls").copy()
    q = np.percentile(st, [50, 90, 99, 99.9, 100])
    print("iteration %d: backward %.1f ms; steps per trajectory mean %.0f, median %.0f, 90%% %.0f, 99%% %.0f, 99.9%% %.0f, max %.0f; sweeps mean %.2f max %d; "
          "balanced time at this kernel's rate would be %.1f ms, the longest trajectory alone is %.2f of the kernel"
          % (it + 1, ms, st.mean(), q[0], q[1], q[2], q[3], q[4], calls.mean(), calls.max(), 0.0, 0.0), end="")
    if prev is not None:
        print("; corr(steps, previous steps) %.2f" % np.corrcoef(prev, st)[0, 1])
    else:
        print()
    prev = st
s.close()
