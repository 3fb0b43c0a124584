import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
from oracle.harness import SYN10_PARAMS, syn10_inputs, Driver, lib_path
B, N = 21, 40
x0, u0 = syn10_inputs(B, N)
fd, b = 1, 7
d = Driver(lib_path("oracle", "synth10hx", fd), N, SYN10_PARAMS, dict(max_iter=1))
assert d.init(x0[b], u0[b]) == 1
d.solve()
xn, un = d.traj(0); cost = d.scalars()["cost"]; lam = d.scalars()["lambda"]
assert d.calc_derivs() == 1
rec, fin = d.derivs()
d.set_lambda(lam); rc = d.back_pass(); l, L = d.gains(); sc = d.scalars()
acc = d.line_search(0); idx = d.log_linesearch(0); s2 = d.scalars(); xc, uc = d.traj(1)
print("oracle: lambda", lam, "bp rc", rc, "accept", acc, "alpha idx", idx, "new cost", s2["new_cost"])
d.close()
for strict in (True, False):
    s = ilqg.BatchSolver("synth10hx", fd, batch=1, n_hor=N, params=SYN10_PARAMS, opts=dict(ls_split=0), strict=strict)
    s.init(x0[b:b+1], u0[b:b+1])
    s.set_x(xn[None]); s.set_u(un[None]); s.set_scalar("cost", cost); s.set_scalar("lambda", lam)
    s.back_pass(fused=True)
    gl, gL = s.gains()
    print("strict", strict, "transient pass: calls", s.ints("bp_calls")[0], "lambda", s.scalar("lambda")[0], "l diff", np.abs(gl[0]-l).max(), "L diff", np.abs(gL[0]-L).max())
    s.line_search()
    print("   line search: accepted", s.ints("accepted")[0], "alpha", s.ints("alpha_idx")[0], "new cost", s.scalar("new_cost")[0], "diff", s.scalar("new_cost")[0]-s2["new_cost"],
          "x diff", np.abs(s.x()[0]-xc).max(), "u diff", np.abs(s.u()[0]-uc).max(), "alpha costs", s.scalar("alpha_cost")[0][:4])
    du = np.abs(s.u()[0]-uc); k = np.unravel_index(du.argmax(), du.shape); print("   worst u at step/input", k, s.u()[0][k[0]], uc[k[0]], "x there", xc[k[0]][:2], "limits u0<=", 0.3+xc[k[0]][1]/2, "u2>=", -(0.3+xc[k[0]][0]**2/5))
    s.close()
