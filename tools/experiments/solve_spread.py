import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg
G = np.load(os.path.join(ROOT, "tests/golden/car_solves_fd0.npz"))
B = len(G["rc"])
for strict in (False, True):
    s = ilqg.BatchSolver("carparking", 0, batch=B, n_hor=500, params=ilqg.CAR_PARAMS, opts=dict(max_iter=int(G["max_iter"])), strict=strict)
    s.init(G["x0"], G["u0"]); s.solve()
    cost, x, it = s.scalar("cost"), s.x(), s.ints("iterations")
    np.set_printoptions(precision=2, linewidth=200)
    print("strict", strict)
    print(" iterations gpu", it.tolist()); print(" iterations ref", G["iterations"].tolist()); print(" iterations fma", G["fma_iterations"].tolist())
    print(" rel cost gpu-ref", np.abs(cost / G["cost"] - 1)); print(" rel cost fma-ref", np.abs(G["fma_cost"] / G["cost"] - 1)); print(" rel cost gpu-fma", np.abs(cost / G["fma_cost"] - 1))
    print(" dx gpu-ref", np.abs(x - G["x"]).max(axis=(1, 2))); print(" dx fma-ref", np.abs(G["fma_x"] - G["x"]).max(axis=(1, 2)))
    print(" dx_end gpu-ref", np.abs(x[:, -1] - G["x"][:, -1]).max(axis=1)); print(" dx_end fma-ref", np.abs(G["fma_x"][:, -1] - G["x"][:, -1]).max(axis=1))
    s.close()
