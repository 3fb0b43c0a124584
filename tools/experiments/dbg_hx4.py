import os, sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
from oracle.harness import SYN10_PARAMS, syn10_inputs, Driver, lib_path
B, N, iters = 21, 40, 3
x0, u0 = syn10_inputs(B, N)
ref = []
for b in range(B):
    d = Driver(lib_path("oracle", "synth10hx", 1), N, SYN10_PARAMS, dict(max_iter=iters))
    assert d.init(x0[b], u0[b]) == 1
    d.solve(); ref.append(d.scalars()["cost"]); d.close()
ref = np.array(ref)
def run(tag, opts, env=None, batch=B, fused=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    s = ilqg.BatchSolver("synth10hx", 1, batch=batch, n_hor=N, params=SYN10_PARAMS, opts=dict(max_iter=iters, **opts), strict=True)
    s.init(x0[:batch], u0[:batch]); s.iterate(iters)
    d = np.abs(s.scalar("cost") - ref[:batch]); print("%-40s worst cost diff %.3e at b=%d; bad: %s" % (tag, d.max(), d.argmax(), np.nonzero(d > 1e-9)[0].tolist()))
    s.close()
    for k in (env or {}): del os.environ[k]
run("default", {})
run("ls_keep=1", dict(ls_keep=1))
run("ls_keep=0", dict(ls_keep=0))
run("ls_split=0", dict(ls_split=0))
run("ls_split=0 ls_keep=0", dict(ls_split=0, ls_keep=0))
run("no rollout parts", {}, dict(ILQG_NO_ROLLOUT_PARTS="1"))
run("no dma", {}, dict(ILQG_NO_DMA="1"))
run("batch 8", {}, batch=8)
run("batch 16", {}, batch=16)
