"""How much work does the lane mapping lose to divergence inside the box QP?  Runs the CPU oracle on one
wavefront's worth of trajectories (64) for the benchmark window and logs, per box-QP call, the number of
outer iterations, factorisations and Armijo trials; prints per-lane means and per-wavefront maxima (what
a wavefront of 64 lanes pays)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import synth
from oracle.harness import CAR_PARAMS, Driver, lib_path

B, N, K = 64, 500, int(sys.argv[1]) if len(sys.argv) > 1 else 20
x0, u0 = synth.car_batch(B, N)
path = lib_path("oracle", full_ddp=0)
lib = C.CDLL(path)
cap = N * (K + 8) * 4
stats = np.zeros((B, K, N, 3), dtype=np.int32)
for b in range(B):
    d = Driver(path, N, CAR_PARAMS, dict(max_iter=K))
    log = np.zeros((cap, 3), dtype=np.int32)
    C.c_void_p.in_dll(d.lib, "ilqg_oracle_boxqp_log").value = log.ctypes.data
    C.c_long.in_dll(d.lib, "ilqg_oracle_boxqp_log_cap").value = cap
    C.c_long.in_dll(d.lib, "ilqg_oracle_boxqp_log_n").value = 0
    assert d.init(x0[b], u0[b]) == 1
    d.solve()
    n = C.c_long.in_dll(d.lib, "ilqg_oracle_boxqp_log_n").value
    C.c_void_p.in_dll(d.lib, "ilqg_oracle_boxqp_log").value = None
    assert n == N * K, (n, N * K)   # no back-pass retries for CarParking FULL_DDP=0
    stats[b] = log[:n].reshape(K, N, 3)
    d.close()
names = ["outer iterations", "factorisations", "Armijo trials"]
for j, nm in enumerate(names):
    s = stats[..., j]
    lane_mean = s.mean()
    wave_max = s.max(axis=0)          # per (iteration, step): what the slowest of 64 lanes needs
    print("%-18s per lane mean %.3f | wavefront (max over 64 lanes) mean %.3f, p50 %d, p90 %d, max %d"
          % (nm, lane_mean, wave_max.mean(), np.percentile(wave_max, 50), np.percentile(wave_max, 90), wave_max.max()))
for it in (0, 4, 9, 14, 19):
    if it < K:
        print("iteration %2d: Armijo trials lane mean %.2f, wavefront max mean %.2f; outer its lane %.2f wave %.2f"
              % (it + 1, stats[:, it, :, 2].mean(), stats[:, it, :, 2].max(axis=0).mean(),
                 stats[:, it, :, 0].mean(), stats[:, it, :, 0].max(axis=0).mean()))
# Armijo trials inside ONE outer iteration cannot be separated here; the distribution over calls:
t = stats[..., 2].ravel()
print("Armijo trials per call histogram:", np.bincount(np.minimum(t, 20))[:21])
