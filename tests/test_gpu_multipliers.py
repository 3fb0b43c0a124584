"""GPU parity tests of the augmented-Lagrangian path: problems with hfe / hli constraints (the reference's
Brachistochrone demos, examples/Brachistochrone/) through the batch C-ABI against the golden fixtures recorded
from the reference and against the CPU oracle.

The Brachistochrone callbacks use nothing but +, -, *, / and sqrt, all IEEE-exact on the device, so the
-ffp-contract=off build must reproduce the reference BIT FOR BIT over a whole solve, multipliers and penalty
weights included; the product build (FMA contraction) is held to the tolerances of test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import golden
from oracle.harness import Driver, almix_case, brachi_case, brachi_hli_case, lib_path

pytestmark = pytest.mark.gpu

CASES = [("fe5_", "brachi", brachi_case(5)), ("fe500_", "brachi", brachi_case(500)),
         ("li500_", "brachi_hli", brachi_hli_case(500))]


def close(a, b, tol=1e-10):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))))


@pytest.fixture(scope="module")
def ilqg():
    import __graft_entry__ as g
    g.build_for_tests()
    from ddp_generator_amd import ilqg as m
    if m.Problem("brachi", 0).device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return m


def solver(ilqg, problem, case, batch=1, strict=False, **extra):
    params, opts, x0, u0 = case
    s = ilqg.BatchSolver(problem, 0, batch=batch, n_hor=len(u0), params=params, opts=dict(opts, **extra), strict=strict)
    return s


@pytest.mark.parametrize("tag,problem,case", CASES)
def test_solve_bit_exact_without_fma(ilqg, tag, problem, case):
    g = golden("brachi.npz")
    _, opts, x0, u0 = case
    s = solver(ilqg, problem, case, strict=True)
    assert s.multiplier_dims() == (g[tag + "mul"].shape[1], g[tag + "mul_fin"].shape[0])
    s.init(x0[None], u0[None])
    assert s.scalar("cost")[0] == g[tag + "init_cost"]  # rolled out with zero penalty weights (iLQG_mex.c:23,116)
    its = int(g[tag + "iterations"])
    for it in range(its):
        s.iterate(1)
        # the trace holds the state at the end of the line search of every iteration
        assert s.scalar("new_cost")[0] == g[tag + "trace_new_cost"][it], it
        assert s.ints("alpha_idx")[0] == g[tag + "trace_alpha_idx"][it], it
        assert s.scalar("g_norm")[0] == g[tag + "trace_g_norm"][it], it
    s.solve()  # the exit test that ends the solve (gradient test at the start of the next iteration, or none)
    assert s.ints("iterations")[0] == its
    assert s.success()[0] == bool(g[tag + "rc"])
    assert s.scalar("cost")[0] == g[tag + "cost"]
    assert np.array_equal(s.x()[0], g[tag + "x"]) and np.array_equal(s.u()[0], g[tag + "u"])
    run, fin = s.multipliers()
    assert np.array_equal(run[0], g[tag + "mul"]) and np.array_equal(fin[0], g[tag + "mul_fin"])
    assert s.scalar("w_pen_l")[0] == g[tag + "w_pen"][0] and s.scalar("w_pen_f")[0] == g[tag + "w_pen"][1]
    s.close()


@pytest.mark.parametrize("tag,problem,case", CASES)
def test_solve_product_build(ilqg, tag, problem, case):
    g = golden("brachi.npz")
    _, opts, x0, u0 = case
    s = solver(ilqg, problem, case)
    s.init(x0[None], u0[None])
    assert close(s.scalar("cost")[0], g[tag + "init_cost"])
    s.solve()
    assert s.ints("iterations")[0] == int(g[tag + "iterations"])
    assert s.success()[0] == bool(g[tag + "rc"])
    assert close(s.scalar("cost")[0], g[tag + "cost"], 1e-6)
    assert np.abs(s.x()[0] - g[tag + "x"]).max() < 1e-4
    assert s.scalar("w_pen_l")[0] == g[tag + "w_pen"][0] and s.scalar("w_pen_f")[0] == g[tag + "w_pen"][1]
    run, fin = s.multipliers()
    assert close(fin[0], g[tag + "mul_fin"], 1e-5)
    s.close()


@pytest.mark.parametrize("compact", [0, 8])
def test_batch_of_different_starts_against_the_oracle(ilqg, oracle_built, compact):
    """every trajectory carries its own multipliers and penalty weights: a batch that spans several wavefronts,
    solved in lock step (strict build: bitwise) against one oracle solve per trajectory.  compact = 8: the same with
    finished trajectories retired — the live ones gathered into smaller contexts, multipliers, weights and stored
    derivative records moving with them (ilqg_batch_solve, option "compact") — every one of the 150 solves still the
    oracle's, bit for bit"""
    B, n = 150, 60
    params, opts, x0, u0 = brachi_hli_case(n)
    rng = np.random.default_rng(11)
    x0s = -10.0 ** rng.uniform(-16, -1, (B, 1))
    u0s = -np.ones((B, n, 1)) * rng.uniform(0.3, 2.0, (B, 1, 1)) + 0.05 * rng.standard_normal((B, n, 1))
    s = ilqg.BatchSolver("brachi_hli", 0, batch=B, n_hor=n, params=params, opts=dict(opts, compact=compact), strict=True)
    s.init(x0s, u0s)
    s.solve()
    assert (s.solve_trace()[3] >= 1) == (compact > 0)
    cost, iters, ok = s.scalar("cost"), s.ints("iterations"), s.success()
    x, (run, fin), wl, wf = s.x(), s.multipliers(), s.scalar("w_pen_l"), s.scalar("w_pen_f")
    s.close()
    seen = set()
    for b in range(B):
        d = Driver(lib_path("oracle", "brachi_hli", 0), n, params, opts)
        assert d.init(x0s[b], u0s[b]) == 1
        rc = d.solve()
        el, fn, w = d.multipliers()
        assert (rc == 1) == bool(ok[b]) and int(d.scalars()["iterations"]) == iters[b], b
        assert d.scalars()["cost"] == cost[b], b
        assert np.array_equal(d.traj(0)[0], x[b]), b
        assert np.array_equal(el, run[b]) and np.array_equal(fn, fin[b]), b
        assert (wl[b], wf[b]) == w, b
        seen.add(w)
        d.close()
    assert len(seen) > 1  # the penalty-weight schedules differ between trajectories


@pytest.mark.parametrize("fd", [0, 1])
def test_all_constraint_kinds_bit_exact_without_fma(ilqg, fd):
    """hle, hli, hfe, hfi at once, arrays of multipliers, a clamped input, a rejected first iteration (weights raised
    by w_pen_fact2 and cost re-swept, iLQG.c:345-349): iteration by iteration against the reference's fixture"""
    check_almix_strict(ilqg, "almix", fd)


def check_almix_strict(ilqg, problem, fd):
    g = golden("almix.npz")
    tag = "fd%d_" % fd
    params, opts, x0, u0 = almix_case()
    s = ilqg.BatchSolver(problem, fd, batch=1, n_hor=len(u0), params=params, opts=opts, strict=True)
    assert s.multiplier_dims() == (g[tag + "mul"].shape[1], g[tag + "mul_fin"].shape[0]) == (6, 6)
    s.init(x0[None], u0[None])
    assert s.scalar("cost")[0] == g[tag + "init_cost"]
    its = int(g[tag + "iterations"])
    for it in range(its):
        s.iterate(1)
        assert s.scalar("new_cost")[0] == g[tag + "trace_new_cost"][it], it
        assert s.ints("alpha_idx")[0] == g[tag + "trace_alpha_idx"][it], it
        assert s.scalar("g_norm")[0] == g[tag + "trace_g_norm"][it], it
    s.solve()
    assert s.ints("iterations")[0] == its and s.success()[0] == bool(g[tag + "rc"])
    assert s.scalar("cost")[0] == g[tag + "cost"]
    assert np.array_equal(s.x()[0], g[tag + "x"]) and np.array_equal(s.u()[0], g[tag + "u"])
    run, fin = s.multipliers()
    assert np.array_equal(run[0], g[tag + "mul"]) and np.array_equal(fin[0], g[tag + "mul_fin"])
    assert s.scalar("w_pen_l")[0] == g[tag + "w_pen"][0] and s.scalar("w_pen_f")[0] == g[tag + "w_pen"][1]
    s.close()


@pytest.mark.parametrize("fd", [0, 1])
def test_all_constraint_kinds_batch_against_the_oracle(ilqg, oracle_built, fd):
    """product build, 70 different starts in lock step: same iteration counts and exits as the oracle, costs and
    trajectories within the full-solve tolerances of test_gpu_parity.py"""
    B = 70
    params, opts, x0, u0 = almix_case(batch=B)
    s = ilqg.BatchSolver("almix", fd, batch=B, n_hor=u0.shape[1], params=params, opts=opts)
    s.init(x0, u0)
    s.solve()
    cost, iters, ok, x = s.scalar("cost"), s.ints("iterations"), s.success(), s.x()
    wl = s.scalar("w_pen_l")
    s.close()
    same = 0
    for b in range(B):
        d = Driver(lib_path("oracle", "almix", fd), u0.shape[1], params, opts)
        assert d.init(x0[b], u0[b]) == 1
        rc = d.solve()
        if int(d.scalars()["iterations"]) == iters[b]:
            same += 1
            assert (rc == 1) == bool(ok[b]), b
            assert close(cost[b], d.scalars()["cost"], 1e-6), b
            assert np.abs(d.traj(0)[0] - x[b]).max() < 1e-4, b
            assert d.multipliers()[2][0] == wl[b], b
        else:  # a tolerance exit taken one iteration earlier or later: same optimum
            assert close(cost[b], d.scalars()["cost"], 1e-5), b
        d.close()
    assert same >= B - 3, same


def test_dropin_line_search_uses_the_callers_multipliers(ilqg, oracle_built):
    """the drop-in iLQG() of the product library (host loop + device back_pass / line_search) on a problem with
    multipliers: same iterations and result as the oracle's"""
    import ctypes as C
    params, opts, x0, u0 = brachi_case(50)
    out = []
    for path in (lib_path("oracle", "brachi", 0), "hip"):
        if path == "hip":
            import os
            path = os.path.join(os.path.dirname(lib_path("oracle")), "libdrv_brachi_fd0_hip.so")
        d = Driver(path, 50, params, opts)
        assert d.init(x0, u0) == 1
        rc = d.solve()
        out.append((rc, d.scalars(), d.traj(0)[0], d.multipliers()))
        d.close()
    a, b = out
    assert a[0] == b[0] and a[1]["iterations"] == b[1]["iterations"]
    assert close(b[1]["cost"], a[1]["cost"], 1e-6) and np.abs(a[2] - b[2]).max() < 1e-4
    assert a[3][2] == b[3][2]  # penalty weights


def test_multipliers_can_be_read_and_set(ilqg):
    """ilqg_batch_get/set_multipliers: member-by-member round trip, and a solve that starts from the multipliers
    and weights another solve ended with needs fewer iterations than the cold one"""
    params, opts, x0, u0 = brachi_case(50)
    a = solver(ilqg, "brachi", brachi_case(50))
    a.init(x0[None], u0[None])
    run0, fin0 = a.multipliers()
    assert run0.shape == (1, 50, 0) and fin0.shape == (1, 2)
    # mu_fe = 0 (init_multipliers); last_hfe = the constraint value of the initial roll-out (update_multipliers(o, 1))
    assert fin0[0, 0] == 0.0 and fin0[0, 1] == a.x()[0, -1, 0] - params["yf"][0]
    a.solve()
    run1, fin1 = a.multipliers()
    assert fin1[0, 0] != 0.0
    cold = int(a.ints("iterations")[0])
    xa, ua = a.x(), a.u()
    b = solver(ilqg, "brachi", brachi_case(50))
    b.init(x0[None], ua)                                 # the solved controls ...
    b.set_multipliers(run1, fin1)                        # ... and the multipliers they were found with
    r2, f2 = b.multipliers()
    assert np.array_equal(f2, fin1)
    b.solve()
    assert int(b.ints("iterations")[0]) < cold
    assert abs(b.x()[0, -1, 0] - xa[0, -1, 0]) < 1e-5
    a.close(); b.close()


@pytest.mark.parametrize("problem,case", [("brachi_hli", brachi_hli_case(500)), ("almix", almix_case())])
def test_fused_and_stored_derivatives_agree(ilqg, problem, case):
    """derivatives evaluated inside the backward kernel (the default) or kept as records in HBM: identical solves.
    The fused kernel re-evaluates them in every sweep, also after a rejected step that raised the penalty weights
    (almix: first iteration), and must then use the weights of the last accepted step, as the reference's kept
    derivatives do (iLQG.c:345-349)."""
    params, opts, x0, u0 = case
    out = []
    for fuse in (1, 0):
        s = ilqg.BatchSolver(problem, 0, batch=1, n_hor=len(u0), params=params, opts=dict(opts, fuse_derivs=fuse), strict=True)
        s.init(x0[None], u0[None])
        hist = []
        for _ in range(opts["max_iter"]):
            s.iterate(1)
            hist.append((s.scalar("cost")[0], s.scalar("new_cost")[0], int(s.ints("alpha_idx")[0]), s.scalar("w_pen_l")[0],
                         s.scalar("g_norm")[0], int(s.ints("status")[0])))
        out.append((hist, s.x(), s.multipliers()))
        s.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2][0], out[1][2][0]) and np.array_equal(out[0][2][1], out[1][2][1])
    assert any(h[2] > 8 for h in out[0][0]) or problem != "almix"  # the rejected iteration is part of it


def test_multipliers_across_stream_groups(ilqg):
    """the batch split into groups on separate streams (ragged sizes): per-trajectory multipliers and weights land
    where they belong — bit-identical to one group, reads and writes of the multiplier arrays included"""
    B = 300  # groups are whole tiles of 64: 128 + 128 + 44
    params, opts, x0, u0 = almix_case(batch=B)
    ref = None
    for groups in (1, 3):
        s = ilqg.BatchSolver("almix", 1, batch=B, n_hor=u0.shape[1], params=params, opts=dict(opts, max_iter=12), groups=groups)
        assert s.groups() == groups
        s.init(x0, u0)
        s.iterate(6)
        run, fin = s.multipliers()
        s.set_multipliers(run, fin)  # write back what was read: a no-op if the group offsets are right
        s.iterate(6)
        run, fin = s.multipliers()
        out = (s.scalar("cost"), s.x(), run, fin, s.scalar("w_pen_l"), s.scalar("w_pen_f"), s.ints("iterations"))
        s.close()
        if ref is None:
            ref = out
        else:
            for a, r in zip(out, ref):
                assert np.array_equal(a, r)
    assert np.ptp(ref[3][:, 0]) > 0  # the trajectories differ, so a misplaced group would show


def test_wave_mapping_with_multipliers(ilqg):
    """the one-wavefront-per-trajectory build of the hli problem (it re-evaluates derivatives chunk by chunk in every
    iteration) against the reference's fixture"""
    g = golden("brachi.npz")
    tag = "li500_"
    params, opts, x0, u0 = brachi_hli_case(500)
    s = ilqg.BatchSolver("brachi_hli", 0, batch=3, n_hor=500, params=params, opts=opts, strict="wave")
    assert s.problem.wave_mapping
    s.init(np.repeat(x0[None], 3, axis=0), np.repeat(u0[None], 3, axis=0))
    assert close(s.scalar("cost"), g[tag + "init_cost"])
    s.solve()
    assert np.all(s.ints("iterations") == int(g[tag + "iterations"]))
    assert close(s.scalar("cost"), g[tag + "cost"], 1e-6)
    assert np.abs(s.x() - g[tag + "x"]).max() < 1e-4
    assert np.all(s.scalar("w_pen_l") == g[tag + "w_pen"][0])
    s.close()
