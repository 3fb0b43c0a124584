"""The CPU restatement (oracle/) against the golden fixtures recorded from the
reference's own sources (tests/golden/make_goldens.py).  Both are compiled with
-ffp-contract=off and the restatement keeps the reference's operation order, so
every comparison here is EXACT (bitwise) — that is what "pinned" means in the
oracle's header."""
import numpy as np
import pytest

from conftest import golden
from oracle.harness import (CAR_PARAMS, HX_N, HX_PARAMS, SYN_PARAMS_TIGHT, SYNP_PARAMS_TIGHT, Driver, Kernels, almix_case, brachi_case, brachi_hli_case,
                            lib_path)


@pytest.fixture(scope="module")
def K(oracle_built):
    return Kernels(lib_path("oracle", full_ddp=0))


def tri(n):
    return n * (n + 1) // 2


def test_cholesky_and_inverse(K):
    g = golden("kernels.npz")
    for n, A, ok, U, inv in zip(g["chol_n"], g["chol_A"], g["chol_ok"], g["chol_U"], g["chol_inv"]):
        t = tri(n)
        ok2, U2 = K.cholesky(A[:t].copy(), int(n))
        assert ok2 == ok
        if ok:
            assert np.array_equal(U2, U[:t])
            assert np.array_equal(K.cholesky_inv(U2, int(n)), inv[:t])


def test_boxqp_every_return_code(K):
    g = golden("kernels.npz")
    seen = set()
    for i in range(len(g["qp_n"])):
        n = int(g["qp_n"][i]); t = tri(n)
        r = K.boxqp(g["qp_H"][i][:t], g["qp_g"][i][:n], g["qp_lo"][i][:n], g["qp_hi"][i][:n], g["qp_x0"][i][:n])
        assert r["rc"] == g["qp_rc"][i]
        seen.add(r["rc"])
        assert np.array_equal(r["x"], g["qp_x"][i][:n])
        if r["rc"] != -1 or True:
            assert np.array_equal(r["clamp"], g["qp_clamp"][i][:n])
            assert r["n_free"] == g["qp_nfree"][i]
        if r["rc"] >= 1 and r["rc"] != 6:
            nf = r["n_free"]
            assert np.array_equal(r["invH"][:tri(nf)], g["qp_invH"][i][:tri(nf)])
    assert seen == {-2, -1, 1, 2, 4, 5, 6}  # every return code of boxQP.c, the 100-iteration exit (rc 1) included


@pytest.mark.parametrize("tag,n,m", [("car", 4, 2), ("syn", 16, 8)])
def test_matmult_helpers(K, tag, n, m):
    g = golden("kernels.npz")
    p = lambda k: g["mm_%s_%s" % (tag, k)]
    assert np.array_equal(K.add_mul_vec(p("base_u"), p("vx"), p("fu"), n, m), p("mulvec"))
    assert np.array_equal(K.add_square_tri(p("base_uu"), p("V"), p("fu"), n, m), p("sq_uu"))
    assert np.array_equal(K.add_square_tri(p("base_xx"), p("V"), p("fx"), n, n), p("sq_xx"))
    assert np.array_equal(K.add_mul2_tri(p("base_xu"), p("V"), p("fx"), n, n, p("fu"), n, m), p("mul2"))


@pytest.mark.parametrize("fd", [0, 1])
def test_single_backward_pass_and_line_search(oracle_built, fd):
    check_car_single(lib_path("oracle", full_ddp=fd), fd)


def check_car_single(path, fd):
    """(also run on builds from other problem files of the same problem: tests/test_template_literal.py)"""
    g = golden("car_single_fd%d.npz" % fd)
    d = Driver(path, 500, CAR_PARAMS)
    assert d.init(g["x0"], g["u0"]) == 1
    assert d.scalars()["cost"] == float(g["init_cost"])
    x, u = d.traj(0)
    assert np.array_equal(x, g["x_nom"]) and np.array_equal(u, g["u_nom"])
    assert d.calc_derivs() == 1
    rec, fin = d.derivs()
    assert np.array_equal(rec, g["rec"]) and np.array_equal(fin, g["fin"])
    assert d.back_pass() == int(g["bp_rc"]) == 0
    l, L = d.gains()
    s = d.scalars()
    assert np.array_equal(l, g["l"]) and np.array_equal(L, g["L"])
    assert s["dV0"] == g["dV"][0] and s["dV1"] == g["dV"][1] and s["g_norm"] == float(g["g_norm"])
    for a, c, ok in zip(g["alphas"], g["alpha_cost"], g["alpha_ok"]):
        ok2, c2 = d.forward_pass(a)
        assert ok2 == ok and c2 == c
    assert d.line_search(0) == int(g["ls_accept"])
    assert d.log_linesearch(0) == int(g["ls_index"])
    s = d.scalars()
    assert s["new_cost"] == float(g["new_cost"]) and s["dcost"] == float(g["dcost"]) and s["expected"] == float(g["expected"])
    xc, uc = d.traj(1)
    assert np.array_equal(xc, g["x_cand"]) and np.array_equal(uc, g["u_cand"])

    # forced Cholesky failure path: back_pass reports 1 (back_pass.c:168-171)
    d.set_derivs(g["bad_rec"], g["fin"])
    d.set_lambda(1.0)
    assert d.back_pass() == int(g["bad_rc"]) == 1


@pytest.mark.parametrize("fd", [0, 1])
def test_full_solves(oracle_built, fd):
    g = golden("car_solves_fd%d.npz" % fd)
    for b in range(0, len(g["rc"]), 3):  # every third trajectory keeps the CPU suite short
        d = Driver(lib_path("oracle", full_ddp=fd), 500, CAR_PARAMS, dict(max_iter=int(g["max_iter"])))
        assert d.init(g["x0"][b], g["u0"][b]) == 1
        assert d.solve() == g["rc"][b]
        s = d.scalars()
        assert int(s["iterations"]) == g["iterations"][b]
        assert s["cost"] == g["cost"][b] and s["lambda"] == g["lam"][b] and s["g_norm"] == g["g_norm"][b]
        x, u = d.traj(0)
        assert np.array_equal(x, g["x"][b]) and np.array_equal(u, g["u"][b])
        t = d.trace()
        n = int(g["tr_len"][b])
        for k in ("lambda", "cost", "new_cost", "alpha_idx", "bp_calls"):
            assert np.array_equal(t[k], g["tr_" + k][b][:n]), k
        d.close()


@pytest.mark.parametrize("fd", [0, 1])
def test_state_dependent_limits_problem(oracle_built, fd):
    """problems/hxtest: the constraint-gradient terms of the gains (back_pass.c:186-199)"""
    check_hxtest(lib_path("oracle", "hxtest", fd), fd)


def check_hxtest(path, fd):
    g = golden("hxtest_fd%d.npz" % fd)
    for tag, pre in (("", 0), ("it3_", 3)):
        d = Driver(path, HX_N, HX_PARAMS, dict(max_iter=max(pre, 1)))
        assert d.init(g["x0"][0], g["u0"][0]) == 1
        if pre:
            d.solve()
        assert d.calc_derivs() == 1
        rec, fin = d.derivs()
        assert np.array_equal(rec, g[tag + "rec"]) and np.array_equal(fin, g[tag + "fin"])
        d.set_lambda(float(g[tag + "lam"]))
        assert d.back_pass() == int(g[tag + "bp_rc"])
        l, L = d.gains()
        assert np.array_equal(l, g[tag + "l"]) and np.array_equal(L, g[tag + "L"])
        assert d.line_search(0) == int(g[tag + "ls_accept"]) and d.log_linesearch(0) == int(g[tag + "ls_index"])
        xc, uc = d.traj(1)
        assert np.array_equal(xc, g[tag + "x_cand"]) and np.array_equal(uc, g[tag + "u_cand"])
        d.close()
    for b in range(len(g["solve_rc"])):
        d = Driver(path, HX_N, HX_PARAMS, dict(max_iter=100))
        assert d.init(g["x0"][b], g["u0"][b]) == 1
        assert d.solve() == g["solve_rc"][b]
        assert d.scalars()["cost"] == g["solve_cost"][b] and np.array_equal(d.traj(0)[0], g["solve_x"][b])
        d.close()


@pytest.mark.parametrize("problem,fd", [("synth16x8", 0), ("synth16x8", 1), ("synth16p", 1)])
def test_synthetic_16x8_problem(oracle_built, problem, fd):
    """n=16, m=8 with dense second derivatives (BASELINE config 5), short horizon; synth16p: the variant with pairwise
    state products in the nonlinearity, whose tensors do not factor (problems/defs/synth16p.py)"""
    g = golden("%s_fd%d.npz" % (problem, fd))
    N = int(g["n_hor"])
    params = SYNP_PARAMS_TIGHT if problem == "synth16p" else SYN_PARAMS_TIGHT
    for tag, pre in (("", 0), ("it3_", 3)):
        d = Driver(lib_path("oracle", problem, fd), N, params, dict(max_iter=max(pre, 1)))
        assert d.init(g["x0"][0], g["u0"][0]) == 1
        if pre:
            d.solve()
        assert d.calc_derivs() == 1
        rec, fin = d.derivs()
        assert np.array_equal(rec, g[tag + "rec"]) and np.array_equal(fin, g[tag + "fin"])
        d.set_lambda(float(g[tag + "lam"]))
        assert d.back_pass() == int(g[tag + "bp_rc"])
        l, L = d.gains()
        assert np.array_equal(l, g[tag + "l"]) and np.array_equal(L, g[tag + "L"])
        assert d.line_search(0) == int(g[tag + "ls_accept"]) and d.log_linesearch(0) == int(g[tag + "ls_index"])
        assert np.array_equal(d.traj(1)[0], g[tag + "x_cand"])
        d.close()
    b = 0
    d = Driver(lib_path("oracle", problem, fd), N, params, dict(max_iter=100))
    assert d.init(g["x0"][b], g["u0"][b]) == 1
    assert d.solve() == g["solve_rc"][b] and d.scalars()["cost"] == g["solve_cost"][b]
    d.close()


def test_regtype2_literal(oracle_built):
    g = golden("car_regtype2.npz")
    for fd in (0, 1):
        d = Driver(lib_path("oracle", full_ddp=fd), 500, CAR_PARAMS, dict(regType=2))
        assert d.init(g["x0"], g["u0"]) == 1 and d.calc_derivs() == 1
        d.set_lambda(1.0)
        assert d.back_pass() == int(g["fd%d_rc" % fd])
        l, L = d.gains()
        assert np.array_equal(l, g["fd%d_l" % fd]) and np.array_equal(L, g["fd%d_L" % fd])
        d.close()


@pytest.mark.parametrize("tag,problem,case", [("fe5_", "brachi", brachi_case(5)), ("fe500_", "brachi", brachi_case(500)),
                                              ("li500_", "brachi_hli", brachi_hli_case(500))])
def test_multiplier_problems(oracle_built, tag, problem, case):
    """augmented-Lagrangian path (update_multipliers, penalty-weight schedule, cost re-sweeps, iLQG.c:233-237,
    337-349) on the reference's Brachistochrone demos: every iteration's scalars, the final trajectory,
    multipliers and penalty weights"""
    g = golden("brachi.npz")
    params, opts, x0, u0 = case
    d = Driver(lib_path("oracle", problem, 0), len(u0), params, opts)
    assert d.init(x0, u0) == 1
    assert d.scalars()["cost"] == g[tag + "init_cost"]
    assert d.solve() == int(g[tag + "rc"])
    x, u = d.traj(0)
    el, fin, w = d.multipliers()
    assert np.array_equal(x, g[tag + "x"]) and np.array_equal(u, g[tag + "u"])
    assert np.array_equal(el, g[tag + "mul"]) and np.array_equal(fin, g[tag + "mul_fin"])
    assert np.array_equal(np.array(w), g[tag + "w_pen"])
    assert d.scalars()["cost"] == g[tag + "cost"] and int(d.scalars()["iterations"]) == int(g[tag + "iterations"])
    for k, v in d.trace().items():
        assert np.array_equal(v, g[tag + "trace_" + k]), k
    d.close()


@pytest.mark.parametrize("fd", [0, 1])
def test_all_constraint_kinds(oracle_built, fd):
    """hle, hli, hfe, hfi together with a clamped input and a rejected first iteration (problems/defs/almix.py)"""
    check_almix(lib_path("oracle", "almix", fd), fd)


def check_almix(path, fd):
    g = golden("almix.npz")
    tag = "fd%d_" % fd
    params, opts, x0, u0 = almix_case()
    d = Driver(path, len(u0), params, opts)
    assert d.init(x0, u0) == 1
    assert d.scalars()["cost"] == g[tag + "init_cost"]
    assert d.solve() == int(g[tag + "rc"])
    x, u = d.traj(0)
    el, fin, w = d.multipliers()
    assert np.array_equal(x, g[tag + "x"]) and np.array_equal(u, g[tag + "u"])
    assert np.array_equal(el, g[tag + "mul"]) and np.array_equal(fin, g[tag + "mul_fin"])
    assert np.array_equal(np.array(w), g[tag + "w_pen"])
    assert d.scalars()["cost"] == g[tag + "cost"] and int(d.scalars()["iterations"]) == int(g[tag + "iterations"])
    for k, v in d.trace().items():
        assert np.array_equal(v, g[tag + "trace_" + k]), k
    # every kind did something: non-trivial multipliers of all four kinds, an input on its limit
    assert np.abs(el[:, 0]).max() > 1e-3 and np.abs(el[:, 2] - 1.0).max() > 1e-3   # mu_le, mu_li
    assert np.abs(fin[0]) > 1e-3 and np.abs(fin[4] - 1.0) > 1e-6                 # mu_fe, mu_fi
    assert np.isclose(u[:, 0].max(), 1.1)
    d.close()


def test_options_follow_reference_rules(oracle_built):
    """setOptParam: same keys, validation and messages as reference iLQG.c:91-216"""
    d = Driver(lib_path("oracle"), 10, CAR_PARAMS)
    assert d.set_opt("tolFun", 1e-6) is None
    assert d.set_opt("tolFun", 0.0) == "parameter must be positive"
    assert d.set_opt("tolFun", [1.0, 2.0]) == "parameter must be scalar"
    assert d.set_opt("lambdaFactor", 0.5) == "parameter must be > 1"
    assert d.set_opt("regType", 3) == "parameter must be in range [1..2]"
    assert d.set_opt("zMin", 1.0) == "parameter must be in range [0..1)"
    assert d.set_opt("debug_level", 7) == "parameter must be in range [0..6]"
    assert d.set_opt("alpha", [1.0, 0.5, 0.5]) == "all alpha must be monotonically decreasing"
    assert d.set_opt("alpha", [1.5]) == "all alpha must be in the range [1.0..0.0)"
    assert d.set_opt("w_pen_init", 1.0) == "no such parameter"  # the stale key of testBrachi.m:13
    assert d.set_opt("alpha", [1.0, 0.5, 0.25]) is None
