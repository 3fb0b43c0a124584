import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
from oracle.harness import Driver, lib_path, SYN_PARAMS_TIGHT
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import syn_inputs
N = 32
B, iters = 9, int(sys.argv[1]) if len(sys.argv) > 1 else 4
x0, u0 = syn_inputs(B, N, first=200)
ref = []
for b in range(B):
    d = Driver(lib_path("oracle", "synth16x8", 1), N, SYN_PARAMS_TIGHT, dict(max_iter=iters))
    assert d.init(x0[b], u0[b]) == 1
    d.solve()
    ref.append(d.scalars()["cost"])
    d.close()
ref = np.array(ref)
for strict in (False, True):
    for fuse in (0, 1):
        s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=SYN_PARAMS_TIGHT, opts=dict(max_iter=iters, fuse_derivs=fuse), strict=strict)
        s.init(x0, u0)
        s.iterate(iters)
        c = s.scalar("cost")
        print("strict", strict, "fuse", fuse, "max rel dev from oracle %.3g" % np.max(np.abs(c - ref) / ref), "bp_calls", s.ints("bp_calls"))
        s.close()
