import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
from oracle.harness import SYN10_PARAMS, syn10_inputs, Driver, lib_path
B, N = 5, 40
x0, u0 = syn10_inputs(B, N)
for fd in (0, 1):
    for strict in (True,):
        # state after 2 oracle iterations (limits active), then one single pass compared stage by stage
        for b in (0, 1):
            d = Driver(lib_path("oracle", "synth10hx", fd), N, SYN10_PARAMS, dict(max_iter=2))
            assert d.init(x0[b], u0[b]) == 1
            d.solve()
            xn, un = d.traj(0); cost = d.scalars()["cost"]; lam = d.scalars()["lambda"]
            assert d.calc_derivs() == 1
            rec, fin = d.derivs()
            d.set_lambda(lam); rc = d.back_pass(); l, L = d.gains(); sc = d.scalars()
            d.close()
            s = ilqg.BatchSolver("synth10hx", fd, batch=1, n_hor=N, params=SYN10_PARAMS, opts=dict(ls_split=0), strict=strict)
            s.init(x0[b:b+1], u0[b:b+1])
            s.set_x(xn[None]); s.set_u(un[None]); s.set_scalar("cost", cost)
            s.calc_derivs()
            grec, gfin = s.derivs()
            dr = np.abs(grec[0] - rec)
            print("fd", fd, "b", b, "records max diff", dr.max(), "at column", np.unravel_index(dr.argmax(), dr.shape), "fin", np.abs(gfin[0]-fin).max(), "rec width", rec.shape)
            s.set_scalar("lambda", lam)
            s.back_pass(single_sweep=True)
            gl, gL = s.gains()
            print("   stored-record pass: rc", s.ints("bp_rc")[0], rc, "l diff", np.abs(gl[0]-l).max(), "L diff", np.abs(gL[0]-L).max())
            s.back_pass(fused=True)
            gl, gL = s.gains()
            print("   fused / transient pass: l diff", np.abs(gl[0]-l).max(), "L diff", np.abs(gL[0]-L).max(), "calls", s.ints("bp_calls")[0])
            s.close()
