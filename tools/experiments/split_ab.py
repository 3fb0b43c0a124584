"""fused backward pass on one wavefront (bw_split=0) against two (bw_split=1): first difference"""
import sys, os, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
pkg = importlib.import_module("ddp-generator_amd")
ilqg, synth = pkg.ilqg, pkg.synth
fd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, N = 33, 500
x0, u0 = synth.car_batch(B, N)
out = []
for split in (0, 1):
    s = ilqg.BatchSolver("carparking", fd, batch=B, n_hor=N, params=ilqg.CAR_PARAMS, opts=dict(max_iter=12, bw_split=split))
    s.init(x0, u0)
    snaps = []
    for it in range(10):
        s.iterate(1)
        l, L = s.gains()
        snaps.append(dict(l=l, L=L, dV0=s.scalar("dV0"), lam=s.scalar("lambda"), bp=s.ints("bp_calls"), rc=s.ints("bp_rc"),
                          st=s.ints("status"), g=s.scalar("g_norm"), cost=s.scalar("cost"), x=s.x()))
    out.append(snaps)
    s.close()
for it in range(10):
    a, b = out[0][it], out[1][it]
    msg = []
    for k in a:
        d = [i for i in range(B) if not np.array_equal(a[k][i], b[k][i], equal_nan=True)]
        if d:
            msg.append("%s %s" % (k, d[:6]))
    print("iteration", it, "bp calls max", a["bp"].max(), "|", "; ".join(msg) if msg else "same")
    if msg:
        i = [i for i in range(B) if not np.array_equal(a["l"][i], b["l"][i])]
        if i:
            i = i[0]
            st = sorted(set(np.argwhere(a["l"][i] != b["l"][i])[:, 0].tolist()))
            print("   trajectory", i, "bp calls", a["bp"][i], b["bp"][i], "rc", a["rc"][i], b["rc"][i], "l differs at steps", st[:4], "...", st[-4:], len(st))
        break
