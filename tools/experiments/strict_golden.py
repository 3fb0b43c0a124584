import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
from oracle.harness import SYN_PARAMS_TIGHT
G = np.load(os.path.join(ROOT, "tests/golden/synth16x8_fd1.npz"))
N = int(G["n_hor"])
for strict in (False, True):
    s = ilqg.BatchSolver("synth16x8", 1, batch=1, n_hor=N, params=SYN_PARAMS_TIGHT, opts=dict(ls_split=0), strict=strict)
    s.init(G["x0"][:1], G["u0"][:1])
    s.set_x(G["x_nom"][None]); s.set_u(G["u_nom"][None]); s.set_scalar("cost", float(G["cost"]))
    s.calc_derivs()
    rec, fin = s.derivs()
    print("strict", strict, "rec dev", np.abs(rec[0] - G["rec"]).max(), "fin dev", np.abs(fin[0] - G["fin"]).max())
    s.set_scalar("lambda", float(G["lam"]))
    s.back_pass(single_sweep=True)
    l, L = s.gains()
    dl = np.abs(l[0] - G["l"]).max(axis=1); dL = np.abs(L[0] - G["L"]).max(axis=1)
    print("   l dev per step (last 6 steps)", dl[-6:], "L dev", dL[-6:], "dV0", s.scalar("dV0")[0] - G["dV"][0])
    s.close()
