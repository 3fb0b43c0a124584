"""Cycle accounting of the quad-mapped backward step (synthetic n=16/m=8 problem, config 5's size).  Needs
    make -C ddp-generator_amd/csrc PROBLEMS=synth16x8 WAVE_PROBLEMS= PLAIN_PROBLEMS= ELEM_LIBS= LIBDIR=../lib_prof OBJDIR=../build_prof EXTRA_HIPFLAGS=-DILQG_PROFILE_SECTIONS FDS=1 STRICT=0
    ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_prof python tools/section_profile_quad.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N, K = int(os.environ.get("B", 16384)), 1000, 2
x0, u0 = synth.synth16_batch(B, N)
s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=K + 2))
s.init(x0, u0)
out = (C.c_ulonglong * 8)()
s.lib.ilqg_dev_section_cycles(out)
names = ["record loads (wait)", "tensor contraction", "Qx, Qu, Vxx fx, Vxx fu", "Qxx", "Qxu, Quu", "box QP", "gains, dV, Vx, Vxx, g_norm"]
for it in range(K):
    s.iterate(1); s.sync()
    s.lib.ilqg_dev_section_cycles(out)
    calls = s.ints("bp_calls").sum()
    v = np.array(list(out), dtype=float)
    steps = max(1.0, v[7])  # steps of all wavefronts (1 to 4 rows at work in each)
    v = v[:7] / steps
    print("iteration %d (%.2f sweeps per trajectory, %.0f wavefront steps per trajectory): %.0f ticks per wavefront step: " % (it + 1, calls / B, steps / B, v.sum())
          + ", ".join("%s %.0f" % (n, x) for n, x in zip(names, v)))
