import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.harness import syn_inputs
from test_gpu_parity import SYN_PARAMS_TIGHT, golden
FD = int(os.environ.get("FD", "0")); gd = golden("synth16x8_fd%d.npz" % FD)
N = int(gd["n_hor"])
B, iters = 5, 4
x0, u0 = syn_inputs(B, N, first=40)
split = int(sys.argv[1]) if len(sys.argv) > 1 else 5
s = ilqg.BatchSolver("synth16x8", FD, batch=B, n_hor=N, params=SYN_PARAMS_TIGHT, opts=dict(max_iter=iters, ls_split=split))
s.init(x0, u0)
mode = sys.argv[2] if len(sys.argv) > 2 else "step"
if mode == "step":
    for it in range(iters):
        s.iterate(1); s.sync()
        print("iter", it, "ok", s.ints("alpha_idx"), s.ints("accepted"), s.ints("slot"), flush=True)
else:
    s.iterate(iters); s.sync()
print(s.scalar("cost"))
