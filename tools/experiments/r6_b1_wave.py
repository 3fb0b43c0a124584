import sys, json
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
g.load_package()
from ddp_generator_amd import ilqg, synth
x0, u0 = synth.car_single()
for variant in (False, "wave"):
    ilqg.solve_single(x0, u0, ilqg.CAR_PARAMS, dict(max_iter=2), strict=variant)
    for it in (20, 100):
        r = ilqg.solve_single(x0, u0, ilqg.CAR_PARAMS, dict(max_iter=it), strict=variant)
        print(variant, it, r["iterations"], round(1e3 * r["seconds"] / max(1, r["iterations"]), 3), "ms/iter cost", r["cost"])
