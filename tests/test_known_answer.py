"""Known answer from the reference's own material: the Brachistochrone demo overlays its solutions with the analytic
cycloid x = a (phi - sin phi), y = a (cos phi - 1), a = 2 (examples/Brachistochrone/testBrachi.m:13-35; end point
(2 pi, -4), g = 9.81).  The same comparison as numbers: the discrete solutions for n = 5, 50, 500 segments converge to
the cycloid and to its descent time T = pi sqrt(a / g); n = 500 is within 2e-4 of T and 0.012 of the curve (the largest
deviation sits in the first segment, where the cycloid starts with a vertical tangent).  This pins the whole solver
(augmented-Lagrangian multipliers included) to something that does not come from any build of the reference.
(The demo's option `w_pen_init` does not exist — setOptParam refuses it, iLQG.c:211-213 — the weights are set through
w_pen_init_l / w_pen_init_f, as oracle/harness.py brachi_case does.)"""
import numpy as np
import pytest
from scipy.optimize import brentq

from oracle.harness import Driver, brachi_case, lib_path

A, G = 2.0, 9.81
T_CYCLOID = np.pi * np.sqrt(A / G)


def cycloid_y(x):
    out = []
    for xi in x:
        phi = 0.0 if xi <= 0 else brentq(lambda p: A * (p - np.sin(p)) - xi, 0.0, np.pi + 1e-9)
        out.append(A * (np.cos(phi) - 1.0))
    return np.array(out)


def check_against_cycloid(results):
    """results: {n: (cost, y[n+1])}"""
    err_t, err_y = {}, {}
    for n, (cost, y) in results.items():
        xs = np.linspace(0.0, 2.0 * np.pi, n + 1)
        err_t[n] = cost / T_CYCLOID - 1.0
        err_y[n] = np.abs(y - cycloid_y(xs)).max()
        assert abs(y[-1] + 4.0) < 1e-6  # the terminal equality constraint y_N = yf
        assert err_t[n] > 0  # a polygon is slower than the optimal curve
    assert err_t[500] < 2e-4 and err_y[500] < 0.012, (err_t, err_y)
    assert err_t[5] > 5 * err_t[50] > 25 * err_t[500]   # first order in the segment length
    assert err_y[5] > 3 * err_y[50] > 9 * err_y[500]
    xs = np.linspace(0.0, 2.0 * np.pi, 501)
    y = results[500][1]
    assert np.abs(y[25:] - cycloid_y(xs)[25:]).max() < 5e-3  # away from the vertical start


def test_brachistochrone_cpu_oracle_converges_to_the_cycloid(oracle_built):
    res = {}
    for n in (5, 50, 500):
        params, opts, x0, u0 = brachi_case(n)
        d = Driver(lib_path("oracle", "brachi", 0), n, params, opts)
        assert d.init(x0, u0) == 1
        assert d.solve() == 1
        res[n] = (d.scalars()["cost"], d.traj(0)[0][:, 0])
        d.close()
    check_against_cycloid(res)


@pytest.mark.gpu
def test_brachistochrone_gpu_converges_to_the_cycloid():
    from conftest import load_package
    ilqg = load_package().ilqg
    res = {}
    for n in (5, 50, 500):
        params, opts, x0, u0 = brachi_case(n)
        s = ilqg.BatchSolver("brachi", 0, batch=1, n_hor=n, params=params, opts=opts)
        s.init(x0[None], u0[None])
        s.solve()
        assert s.success()[0] == 1
        res[n] = (float(s.scalar("cost")[0]), s.x()[0][:, 0])
        s.close()
    check_against_cycloid(res)
