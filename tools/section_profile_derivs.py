"""Cycle accounting of k_derivs_wave (synthetic n=16/m=8 problem, config 5's size).  Needs
    make -C ddp-generator_amd/csrc PROBLEMS=synth16x8 WAVE_PROBLEMS= PLAIN_PROBLEMS= ELEM_LIBS= LIBDIR=../lib_prof OBJDIR=../build_prof EXTRA_HIPFLAGS=-DILQG_PROFILE_SECTIONS FDS=1 STRICT=0
    ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_prof python tools/section_profile_derivs.py"""
import ctypes as C, os, re, sys
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
def probe_names():
    """the probes in the order a wavefront passes them: kernel code, then the pieces the function file closes with
    ILQG_REC_DONE (products first: k_derivs_wave calls bp_tensor_basis ahead of bp_derivsL_first)"""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "problems", "synth16x8", "iLQG_func.c")).read()
    pieces = re.findall(r"^\s+ILQG_REC_DONE\((\w+), (\d+), (\d+)\)", src, flags=re.M)
    names = ["x, u into the private element", "calcXVariableAux", "calcXUVariableAux", "calcLAuxDeriv",
             "products: sin / cos of the auxiliaries, the 32 products", "  write-out products", "products: rest"]
    for i, (m, a, n) in enumerate(pieces):
        names += ["first derivatives: up to %s[%s..+%s]%s" % (m, a, n, " (with the shared products cs*)" if i == 0 else ""), "  write-out %s[%s..+%s]" % (m, a, n)]
    names += ["first derivatives: rest", "limitsU, limits staged", "  write-out limits", "entries outside the runs"]
    return names


names = probe_names()
waves = B * (N + 1) / 64.0
for it in range(K):
    s.iterate(1); s.sync()
    s.lib.ilqg_dev_derivs_cycles(out)
    v = np.array(list(out), dtype=float) / waves
    print("iteration %d: %.0f ticks per wavefront" % (it + 1, v.sum()))
    for n, x in zip(names, v):
        print("  %-32s %8.0f" % (n, x))
