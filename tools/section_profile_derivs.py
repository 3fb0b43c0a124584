"""Cycle accounting of k_derivs_wave (synthetic n=16/m=8 problem, config 5's size).  Needs
    make -C ddp-generator_amd/csrc PROBLEMS=synth16x8 WAVE_PROBLEMS= PLAIN_PROBLEMS= ELEM_LIBS= LIBDIR=../lib_prof OBJDIR=../build_prof EXTRA_HIPFLAGS=-DILQG_PROFILE_SECTIONS FDS=1 STRICT=0
    ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_prof python tools/section_profile_derivs.py"""
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
out = (C.c_ulonglong * 32)()
s.lib.ilqg_dev_derivs_cycles(out)
names = ["x, u into the record", "calcXVariableAux", "calcXUVariableAux", "calcLAuxDeriv"]
for j, nm in enumerate(["fx 0-63", "fx 64-127", "fx 128-191", "fx 192-255", "fu 0-63", "fu 64-127", "cx", "cu"]):
    names += ["first: up to " + nm, "  write-out " + nm]
names += ["first: rest", "basis: products", "  write-out products", "basis: rest", "limitsU"]
waves = B * (N + 1) / 64.0
for it in range(K):
    s.iterate(1); s.sync()
    s.lib.ilqg_dev_derivs_cycles(out)
    v = np.array(list(out), dtype=float) / waves
    print("iteration %d: %.0f ticks per wavefront" % (it + 1, v.sum()))
    for n, x in zip(names, v):
        print("  %-32s %8.0f" % (n, x))
