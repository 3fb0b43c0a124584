"""Where do the wavefronts of the headline's kernels run while the four stream groups overlap?  A lone k_backward<2> wavefront
issues vector instructions in 67 % of its cycles (profiles/issue.json), a lone k_search<0> wavefront in 60 %: two wavefronts
on one SIMD compete, an empty SIMD next to them is lost.  The workgroup dispatcher places wavefronts; this script logs the
SIMD of every wavefront with its entry and exit time and counts who shared a SIMD with whom.
Needs a -DILQG_WAVE_PLACES build:
    tools/variant.sh places "-DILQG_WAVE_PLACES=1" carparking 0
    ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_places python tools/experiments/wave_places.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
B, N, K = 65536, 500, int(os.environ.get("K", 6))
groups = int(os.environ.get("GROUPS", 0))
x0, u0 = synth.car_batch(B, N)
s = ilqg.BatchSolver("carparking", 0, batch=B, n_hor=N, params=ilqg.CAR_PARAMS, opts=dict(max_iter=K + 12), groups=groups)
s.init(x0, u0)
MAX = 1 << 19
buf = (C.c_ulonglong * (3 * MAX))()
cnt = C.c_int(0)
s.iterate(4); s.sync()
s.lib.ilqg_dev_wave_places(buf, MAX, C.byref(cnt))  # clears the log
t0 = time.perf_counter()
s.iterate(K); s.sync()
dt = time.perf_counter() - t0
s.lib.ilqg_dev_wave_places(buf, MAX, C.byref(cnt))
n = cnt.value
a = np.frombuffer(buf, dtype=np.uint64)[:3 * n].reshape(n, 3).copy()
n_groups = s.groups()
s.close()
kind = (a[:, 0] >> np.uint64(40)).astype(int)
xcc = ((a[:, 0] >> np.uint64(32)) & np.uint64(15)).astype(int)
hw = (a[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
t_in, t_out = a[:, 1].astype(np.int64), a[:, 2].astype(np.int64)
t_in -= t_in.min(); t_out -= a[:, 1].astype(np.int64).min()
print("%d groups, %d iterations in %.1f ms (%.1f it/s); %d wavefronts logged: backward %d, search stage 0 %d, stage 1 %d; clock 100 MHz"
      % (n_groups, K, 1e3 * dt, K / dt, n, (kind == 1).sum(), (kind == 2).sum(), (kind == 3).sum()))
place = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
cuid = place // 4
print("distinct SIMDs seen %d (of 1024), CUs %d, XCCs %d; se values %s sh %s cu %s" % (len(np.unique(place)), len(np.unique(cuid)), len(np.unique(xcc)),
      np.unique(se), np.unique(sh), np.unique(cu)))
dur = (t_out - t_in) / 100.0  # us
for k, name in ((1, "backward"), (2, "search 0"), (3, "search 1")):
    m = kind == k
    if m.any():
        print("%-9s duration us: mean %.0f, 10%% %.0f, median %.0f, 90%% %.0f, max %.0f" % (name, dur[m].mean(), *np.percentile(dur[m], [10, 50, 90, 100])))
# residency per SIMD over time: sample the window on a grid
lo, hi = np.percentile(t_in, 5), np.percentile(t_out, 95)
grid = np.linspace(lo, hi, 400)
ids = np.unique(place)
index = {p: i for i, p in enumerate(ids)}
pi = np.array([index[p] for p in place])
occ = np.zeros((len(grid), len(ids), 4), dtype=np.int16)
for j, t in enumerate(grid):
    live = (t_in <= t) & (t_out > t)
    for k in (1, 2, 3):
        np.add.at(occ[j, :, k], pi[live & (kind == k)], 1)
tot = occ[:, :, 1:].sum(axis=2)
print("wavefronts resident per SIMD (sampled over the steady window, %d SIMDs seen): mean %.2f; share of SIMD-time with 0 / 1 / 2 / 3 / 4+ wavefronts: %s"
      % (len(ids), tot.mean() * len(ids) / 1024.0, " / ".join("%.2f" % ((tot == v).mean() if v < 4 else (tot >= 4).mean()) for v in range(5))))
bw = occ[:, :, 1]
print("backward wavefronts per SIMD: share of SIMD-time with 0 / 1 / 2 / 3+: %s" % " / ".join("%.2f" % ((bw == v).mean() if v < 3 else (bw >= 3).mean()) for v in range(4)))
m = bw > 0
print("a SIMD that holds a backward wavefront holds on average %.2f backward and %.2f search wavefronts; one without holds %.2f search wavefronts"
      % (bw[m].mean(), (tot - bw)[m].mean(), (tot - bw)[~m].mean()))
# per CU
occ_cu = tot.reshape(len(grid), -1)
# duration of a backward wavefront against the company it had (time-averaged wavefronts on its SIMD)
bsel = np.where(kind == 1)[0]
comp = np.zeros(len(bsel))
for q, i in enumerate(bsel):
    same = (place == place[i])
    ov = np.minimum(t_out[same], t_out[i]) - np.maximum(t_in[same], t_in[i])
    comp[q] = np.clip(ov, 0, None).sum() / max(1, t_out[i] - t_in[i]) - 1.0
d = dur[bsel]
print("backward wavefront: company on its SIMD (time-averaged other wavefronts) mean %.2f, 10%% %.2f, 90%% %.2f; corr(duration, company) %.2f"
      % (comp.mean(), *np.percentile(comp, [10, 90]), np.corrcoef(d, comp)[0, 1]))
for lo_c, hi_c in ((0, 0.5), (0.5, 1.0), (1.0, 1.5), (1.5, 2.0), (2.0, 3.0), (3.0, 99)):
    mm = (comp >= lo_c) & (comp < hi_c)
    if mm.any():
        print("   company %.1f-%.1f: %5d wavefronts, mean duration %.0f us" % (lo_c, hi_c, mm.sum(), d[mm].mean()))
