import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
from oracle.harness import SYN10_PARAMS, syn10_inputs, Driver, lib_path
B, N, iters = 21, 40, 5
x0, u0 = syn10_inputs(B, N)
s = ilqg.BatchSolver("synth10hx", 1, batch=B, n_hor=N, params=SYN10_PARAMS, opts=dict(max_iter=iters), strict=True)
s.init(x0, u0)
tr = []
for b in range(B):
    d = Driver(lib_path("oracle", "synth10hx", 1), N, SYN10_PARAMS, dict(max_iter=iters))
    assert d.init(x0[b], u0[b]) == 1
    c0 = d.scalars()["cost"]
    d.solve(); t = d.trace(); t["c0"] = c0; tr.append(t); d.close()
print("init cost diff", np.abs(s.scalar("cost") - np.array([t["c0"] for t in tr])).max())
for it in range(iters):
    s.iterate(1)
    c, a, calls, lam = s.scalar("cost"), s.ints("alpha_idx"), s.ints("bp_calls"), s.scalar("lambda")
    bad = [(b, c[b], tr[b]["new_cost"][it] if it < len(tr[b]["cost"]) else None, int(a[b]), int(tr[b]["alpha_idx"][it]), int(calls[b]), int(tr[b]["bp_calls"][it]))
           for b in range(B) if it < len(tr[b]["alpha_idx"]) and (a[b] != tr[b]["alpha_idx"][it] or calls[b] != tr[b]["bp_calls"][it])]
    x = s.x(); u = s.u()
    dif = [(b, c[b] - tr[b]["new_cost"][it]) for b in range(B) if it < len(tr[b]["new_cost"])]
    worst = max(dif, key=lambda q: abs(q[1]))
    print("iteration", it + 1, "worst cost - oracle new_cost", worst, "accepted", s.ints("accepted")[worst[0]])
    print("iteration", it + 1, "mismatching (b, cost, oracle new_cost, alpha, oracle alpha, calls, oracle calls):", bad[:6])
s.close()
