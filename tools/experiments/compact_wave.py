"""Does `compact` work in the wave mapping (n = 16)?  plain against compacted solve of a small synth16x8 batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.build()
from ddp_generator_amd import ilqg
from oracle.harness import SYN_PARAMS_TIGHT, syn_inputs
for fd in (1, 0):
    for N, B, mi in ((40, 96, 120), (60, 200, 200)):
        x0, u0 = syn_inputs(B, N)
        out = []
        for compact in (0, 8):
            s = ilqg.BatchSolver("synth16x8", fd, batch=B, n_hor=N, params=SYN_PARAMS_TIGHT, opts=dict(max_iter=mi, compact=compact))
            s.init(x0, u0); s.solve()
            l, L = s.gains()
            out.append(dict(x=s.x(), u=s.u(), l=l, L=L, cost=s.scalar("cost").copy(), lam=s.scalar("lambda").copy(), st=s.ints("status").copy(), it=s.ints("iterations").copy(), tr=s.solve_trace()))
            s.close()
        p, c = out
        same = {k: bool(np.array_equal(p[k], c[k])) for k in p if k != "tr"}
        print("fd %d N %d B %d: finished %d of %d, distinct iteration counts %d (min %d max %d), compactions %d, slots %s; identical: %s"
              % (fd, N, B, (p["st"] != 0).sum(), B, len(np.unique(p["it"])), p["it"].min(), p["it"].max(), c["tr"][3], list(c["tr"][2][::max(1, len(c["tr"][2]) // 8)]), same), flush=True)
