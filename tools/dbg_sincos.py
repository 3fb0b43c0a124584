import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpmath, torch
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
mpmath.mp.prec = 2000
x = np.array([7.9e5, 8.0e5, 1e6, 1e7, 1e9, 1e12, 1e15, 1e18, 1e22, 1e30, 1e50, 1e100, 1.9788513965078275e+168, 1e300, np.nan, np.inf])
s, c = ilqg.sincos_batch(x)
ts = torch.sin(torch.tensor(x, device="cuda")).cpu().numpy()
for i, xv in enumerate(x):
    if np.isfinite(xv):
        es = float(mpmath.sin(mpmath.mpf(float(xv))))
    else:
        es = float("nan")
    print("%-12g mine % .17g torch % .17g exact % .17g numpy % .17g" % (xv, s[i], ts[i], es, np.sin(xv)))
