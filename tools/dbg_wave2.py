import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
fd, B, N = int(sys.argv[1]), 1024, 200
x0, u0 = synth.synth16_batch(B, N)
s = ilqg.BatchSolver("synth16x8", fd, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=10))
s.init(x0, u0)
s.iterate(2)
s.sync()
print("done", s.scalar("cost").mean())
