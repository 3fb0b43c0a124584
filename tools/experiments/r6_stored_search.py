"""round 6: the line search of config 5 with stored tensors (scalar roll-outs: a pair without step parts) under the solver's staging
options — ls_split (step sizes of the first stage) and ls_keep (1: second stage beside the re-rolled winners, 0: second stage, then
the winner pass)"""
import sys, time, json
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N = 16384, 1000
x0, u0 = synth.synth16_batch(B, N)
for opts in (dict(), dict(ls_split=0), dict(ls_split=2), dict(ls_split=1, ls_keep=0), dict(ls_split=3)):
    s = ilqg.BatchSolver("synth16x8_plain", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=5, **opts))
    s.init(x0, u0)
    s.iterate(1); s.sync(); s.init(x0, u0)
    s.timing(True); s.sync()
    t0 = time.perf_counter(); s.iterate(2); s.sync(); dt = time.perf_counter() - t0
    busy = s.kernel_busy()
    print(opts, round(2 / dt, 3), "it/s", {k: round(v / 2, 1) for k, v in busy.items() if v > 2}, float(s.scalar("cost").mean()), flush=True)
    s.close()
