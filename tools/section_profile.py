"""Where do the cycles of the fused backward kernel go?  Needs a build with -DILQG_PROFILE_SECTIONS:
    make -C ddp-generator_amd/csrc PROBLEMS=carparking WAVE_PROBLEMS= LIBDIR=../lib_prof OBJDIR=../build_prof EXTRA_HIPFLAGS=-DILQG_PROFILE_SECTIONS
    ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_prof python tools/section_profile.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
K, B = 20, 65536
x0, u0 = synth.car_batch(B, 500)
s = ilqg.BatchSolver("carparking", 0, batch=B, n_hor=500, params=ilqg.CAR_PARAMS, opts=dict(max_iter=K + 2))
s.init(x0, u0)
out = (C.c_ulonglong * 8)()
s.lib.ilqg_dev_section_cycles(out)
names = ["derivs of step k-1 -> loop top + prefetch issue", "Q assembly + regularisation", "box QP: gradient/clamp/search (no factor, no Armijo)",
         "box QP: factorisation + inverse", "box QP: Armijo loop", "gains + value update (+ overlapped derivs issue)", "derivative evaluation tail + exit tests", "loop head"]
for it in range(K):
    s.iterate(1); s.sync()
    s.lib.ilqg_dev_section_cycles(out)
    v = np.array(list(out), dtype=float) / 1024 / 500   # per wavefront and step
    if it in (0, 4, 9, 14, 19):
        print("iteration %2d: %.0f ticks per step:" % (it + 1, v.sum()), ", ".join("%s %.0f" % (n.split(":")[0] if False else n[:28], x) for n, x in zip(names, v)))
