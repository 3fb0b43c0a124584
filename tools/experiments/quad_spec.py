"""Speculative retries of the quad-mapped backward kernel (ILQG_QUAD_SPEC=1) against the plain kernel: results bit for bit on
small batches (many idle rows: every trajectory gets helpers), then the kernel's time at config 5's size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
from oracle.harness import SYN_PARAMS_TIGHT, syn_inputs

def run(spec, B, N, iters, params, x0, u0, timing=False, light=False):
    """light: per-trajectory scalars only (at config 5's size the gains of one iteration are 18 GB of host memory: the next
    iteration's line search carries any difference in them into the scalars)"""
    os.environ["ILQG_QUAD_SPEC"] = "1" if spec else "0"
    s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=params, opts=dict(max_iter=iters + 2))
    s.init(x0, u0)
    out = []
    ms = []
    for it in range(iters):
        if timing: s.timing(True)
        t0 = time.perf_counter()
        s.iterate(1); s.sync()
        dt = time.perf_counter() - t0
        if timing: ms.append((s.kernel_times().get("k_backward", (0, 0.0))[1], 1e3 * dt))
        o = dict(lam=s.scalar("lambda").copy(), dlam=s.scalar("dlambda").copy(), dV0=s.scalar("dV0").copy(), dV1=s.scalar("dV1").copy(), g=s.scalar("g_norm").copy(),
                 cost=s.scalar("cost").copy(), st=s.ints("status").copy(), calls=s.ints("bp_calls").copy(), rc=s.ints("bp_rc").copy())
        if not light:
            l, L = s.gains()
            o.update(l=l.copy(), L=L.copy(), x=s.x().copy())
        out.append(o)
    s.close()
    return out, ms

mode = sys.argv[1] if len(sys.argv) > 1 else "check"
if mode in ("check", "stress"):
    cases = ((40, 40, 6), (300, 60, 6), (1500, 100, 4), (5000, 200, 3)) if mode == "check" else ((16384, 1000, 8), (16384, 1000, 8), (9000, 1000, 6))
    for B, N, iters in cases:
        if mode == "check":
            x0, u0 = syn_inputs(B, N)
            params = SYN_PARAMS_TIGHT
        else:
            x0, u0 = synth.synth16_batch(B, N)
            params = synth.SYNTH16_PARAMS
        a, _ = run(False, B, N, iters, params, x0, u0, light=(mode == "stress"))
        b, _ = run(True, B, N, iters, params, x0, u0, light=(mode == "stress"))
        for it in range(iters):
            bad = [k for k in a[it] if not np.array_equal(a[it][k], b[it][k])]
            print("B %d N %d iteration %d: sweeps mean %.2f max %d; differing fields: %s" % (B, N, it + 1, a[it]["calls"].mean(), a[it]["calls"].max(), bad or "none"), flush=True)
            if bad:
                k = bad[0]; w = np.argwhere(a[it][k] != b[it][k])
                print("   first difference in", k, "at", w[0], a[it][k][tuple(w[0])], b[it][k][tuple(w[0])], "trajectories differing:", len(np.unique(w[:, 0])), "calls there", a[it]["calls"][w[0][0]], b[it]["calls"][w[0][0]])
                sys.exit(1)
elif mode == "time":
    B, N, iters = int(os.environ.get("B", 16384)), 1000, 5
    x0, u0 = synth.synth16_batch(B, N)
    for spec in (False, True, False, True):
        o, ms = run(spec, B, N, iters, synth.SYNTH16_PARAMS, x0, u0, timing=True, light=True)
        print("spec %d: k_backward ms per iteration %s; iteration ms %s; sweeps mean %.2f; cost check %.6f" % (spec, ["%.1f" % m[0] for m in ms], ["%.1f" % m[1] for m in ms],
              o[-1]["calls"].mean(), float(o[-1]["dV0"].sum())), flush=True)

if mode == "stats":
    import ctypes as C
    B, N = int(os.environ.get("B", 16384)), 1000
    x0, u0 = synth.synth16_batch(B, N)
    os.environ["ILQG_QUAD_SPEC"] = "1"
    s = ilqg.BatchSolver("synth16x8", 1, batch=B, n_hor=N, params=synth.SYNTH16_PARAMS, opts=dict(max_iter=8))
    s.init(x0, u0)
    for it in range(3):
        s.iterate(1); s.sync()
        n = C.c_size_t(0)
        cap = B * 19 + 8192 * 4
        buf = (C.c_uint * cap)()
        s.lib.ilqg_dev_spec_words.argtypes = [C.POINTER(C.c_uint), C.c_size_t, C.POINTER(C.c_size_t)]
        s.lib.ilqg_dev_spec_words(buf, cap, C.byref(n))
        w = np.frombuffer(buf, dtype=np.uint32)[:n.value]
        word, done, out, best = w[:B], w[B:2 * B], w[2 * B:18 * B].reshape(B, 16), w[18 * B:19 * B]
        claimed = word & 0xffff
        calls = s.ints("bp_calls")
        kind = out & 3
        ran = (kind != 0).sum(axis=1)
        win = best.astype(np.int64)
        wo = out[np.arange(B), np.minimum(win, 15)]
        by_helper = ((wo >> 2) & 1) == 0
        print("iteration %d: sweeps %.2f; attempts handed out per trajectory mean %.2f (sequential would be %.2f), attempts that ended %.2f; trajectories whose result came out of a row buffer %d of %d; "
              "attempts handed out beyond the winner %d; unknown-but-claimed %d"
              % (it + 1, calls.mean(), claimed.mean(), calls.mean(), ran.mean(), int(by_helper.sum()), B, int((claimed.astype(int) - calls).clip(0).sum()), int((claimed.astype(int) - ran).sum())), flush=True)
    s.close()
