"""duration of the second line-search launch of the n = 16 problem for different step-size tables / splits"""
import sys, os, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
pkg = importlib.import_module("ddp-generator_amd")
ilqg, synth = pkg.ilqg, pkg.synth
B, N = 16384, 1000
x0, u0 = synth.synth16_batch(B, N)
full = list(10.0 ** np.linspace(0, -3, 8))
cases = []
for n in (2, 3, 5, 8):
    cases.append(("%d alphas, no keep" % n, dict(alpha=full[:n], ls_keep=0)))
for n in (2, 8):
    cases.append(("%d alphas, no keep, zMin 0.999" % n, dict(alpha=full[:n], ls_keep=0, zMin=0.999)))
for name, opts in cases:
    s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=6, **opts))
    s.init(x0, u0)
    s.iterate(1)
    s.sync()
    s.timing(True)
    s.iterate(3)
    s.sync()
    t = s.kernel_times()
    print(name, {k: round(v[1] / 3, 1) for k, v in t.items() if v[0] and "rollout" in k}, "accepted idx hist", np.bincount(s.ints("alpha_idx"), minlength=9)[:9])
    s.close()
