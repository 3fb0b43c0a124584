"""The drop-in boundary on every build: the reference-style driver (oracle/driver.c, the call sequence of the MEX entry,
iLQG_mex.c:55-137) linked against the PRODUCT's `back_pass()` / `line_search()` / `iLQG()` (reference iLQG.h:78-88,
back_pass.h:7, line_search.h:6) against the same driver on the CPU oracle — all 6 problems x FULL_DDP 0/1.  This is the
route a Maxima-generated problem takes through `tOptSet`: the caller's `trajEl_t` arrays are packed, the kernels run
with a batch of one, and `l`, `L`, `dV`, `g_norm` / the candidate trajectory are unpacked.  It covers what the batch
tests do not: FULL_DDP = 1 through tOptSet, the sign / hx unpack of `hxtest`, the multiplier structs handed over by
the caller, and the wave mapping's drop-in back_pass() with its private record buffer (synth16x8).

Teacher forced: before every iteration the product-side driver receives the oracle's nominal trajectory, cost,
lambda, multipliers and weights, so every iteration is a single-pass comparison at the single-pass tolerance
(1e-10 relative, flags / return codes / accepted step index exact)."""
import os

import numpy as np
import pytest

from oracle.harness import (CAR_N, CAR_PARAMS, CAR_X0, CONSOLE_CASES, HX_N, HX_PARAMS, SYN_PARAMS_TIGHT, Driver,
                            almix_case, brachi_case, brachi_hli_case, console_of, hx_inputs, lib_path, syn_inputs)

pytestmark = pytest.mark.gpu

TOL = 1e-10
ITERS = 4


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))))


def worst(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def case(problem):
    """(n_hor, params, opts, x0, u0) — small horizons: the drop-in path is one trajectory per call"""
    if problem == "carparking":
        rng = np.random.default_rng(11)
        return CAR_N, CAR_PARAMS, {}, np.array(CAR_X0), 0.1 * rng.standard_normal((CAR_N, 2))
    if problem == "hxtest":
        x0, u0 = hx_inputs(1)
        return HX_N, HX_PARAMS, {}, x0[0], u0[0]
    if problem == "synth16x8":
        x0, u0 = syn_inputs(1, 40)
        return 40, SYN_PARAMS_TIGHT, {}, x0[0], u0[0]
    if problem == "brachi":
        params, opts, x0, u0 = brachi_case(50)
        return 50, params, opts, x0, u0
    if problem == "brachi_hli":
        params, opts, x0, u0 = brachi_hli_case(120)
        return 120, params, opts, x0, u0
    if problem == "almix":
        params, opts, x0, u0 = almix_case()
        return len(u0), params, opts, x0, u0
    raise ValueError(problem)


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build_for_tests()
    from ddp_generator_amd import ilqg as m
    if m.Problem("carparking", 0).device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return True


PROBLEMS = ["carparking", "hxtest", "synth16x8", "brachi", "brachi_hli", "almix"]


def state_after(problem, fd, iterations):
    """what the oracle's iLQG() leaves after `iterations` iterations: nominal trajectory, multipliers, weights,
    lambda and cost — the state the next iteration's stages start from"""
    n_hor, params, opts, x0, u0 = case(problem)
    d = Driver(lib_path("oracle", problem, fd), n_hor, params, dict(opts, max_iter=iterations))
    assert d.init(x0, u0) == 1
    d.solve()
    s = d.scalars()
    out = (d.traj(0), d.multipliers(), s["cost"], s["lambda"])
    d.close()
    return out


def stages(ref, dev, it):
    """one iteration's stages on both drivers from the state they hold; returns whether a step was accepted"""
    assert ref.calc_derivs() == 1 and dev.calc_derivs() == 1
    lam = ref.scalars()["lambda"]
    for attempt in range(12):  # a failed sweep (return 1) raises lambda as iLQG.c:267-275 does
        ra, rb = ref.back_pass(), dev.back_pass()
        assert ra == rb, (it, attempt, ra, rb)
        if ra == 0:
            break
        lam = max(lam * 1.6, 1e-6)
        ref.set_lambda(lam)
        dev.set_lambda(lam)
    assert ra == 0
    (la, La), (lb, Lb) = ref.gains(), dev.gains()
    sa, sb = ref.scalars(), dev.scalars()
    assert close(lb, la) and close(Lb, La), (it, worst(lb, la), worst(Lb, La))
    for key in ("dV0", "dV1", "g_norm"):
        assert close(sb[key], sa[key]), (it, key, sb[key], sa[key])
    # line search on the gains each side computed
    fa, fb = ref.line_search(it), dev.line_search(it)
    assert fa == fb, (it, fa, fb)
    assert ref.log_linesearch(it) == dev.log_linesearch(it), it
    sa, sb = ref.scalars(), dev.scalars()
    if fa:
        for key in ("new_cost", "dcost", "expected"):
            assert close(sb[key], sa[key]), (it, key, sb[key], sa[key])
        (xa, ua), (xb, ub) = ref.traj(1), dev.traj(1)
        assert close(xb, xa) and close(ub, ua), (it, worst(xb, xa), worst(ub, ua))
    return bool(fa)


@pytest.mark.parametrize("fd", [0, 1])
@pytest.mark.parametrize("problem", PROBLEMS)
def test_dropin_stages_match_the_oracle(built, problem, fd):
    n_hor, params, opts, x0, u0 = case(problem)
    hip = os.path.join(os.path.dirname(lib_path("oracle")), "libdrv_%s_fd%d_hip.so" % (problem, fd))
    assert os.path.exists(hip), hip
    ref = Driver(lib_path("oracle", problem, fd), n_hor, params, opts)
    dev = Driver(hip, n_hor, params, opts)
    assert ref.init(x0, u0) == 1 and dev.init(x0, u0) == 1
    # the initial roll-out is the host's generated forward_pass on both sides
    assert ref.scalars()["cost"] == dev.scalars()["cost"]
    accepted = 0
    for it in range(ITERS):
        (x, u), (el, fin, w), cost, lam = state_after(problem, fd, it)
        for d in (ref, dev):
            d.set_state(x, u, cost, lam, w)
            d.set_multipliers(el, fin)
        accepted += stages(ref, dev, it)
    assert accepted >= 2  # (almix: the first iteration is rejected by design)
    ref.close()
    dev.close()


@pytest.mark.parametrize("fd", [0, 1])
@pytest.mark.parametrize("problem", ["carparking", "hxtest", "synth16x8", "almix"])
def test_dropin_solve_matches_the_oracle(built, problem, fd):
    """iLQG(tOptSet*) of the product (host loop around the device stages, reference iLQG.c:224-379) for a few
    iterations, free running: same number of iterations, same accepted step sizes, cost and trajectory within the
    amplification a handful of iterations allows"""
    n_hor, params, opts, x0, u0 = case(problem)
    opts = dict(opts, max_iter=5)
    hip = os.path.join(os.path.dirname(lib_path("oracle")), "libdrv_%s_fd%d_hip.so" % (problem, fd))
    out = []
    for path in (lib_path("oracle", problem, fd), hip):
        d = Driver(path, n_hor, params, opts)
        assert d.init(x0, u0) == 1
        rc = d.solve()
        s = d.scalars()
        out.append((rc, int(s["iterations"]), [d.log_linesearch(i) for i in range(int(s["iterations"]))], s["cost"], d.traj(0)))
        d.close()
    a, b = out
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], (a[:3], b[:3])
    assert close(b[3], a[3], 1e-8), (a[3], b[3])
    assert np.abs(a[4][0] - b[4][0]).max() < 1e-4 and np.abs(a[4][1] - b[4][1]).max() < 1e-4  # full-solve tolerance


def _same_console(got, want):
    """line by line: the same words, and the same numbers to the three digits most of them are printed with"""
    got, want = got.splitlines(), want.splitlines()
    assert len(got) == len(want), (len(got), len(want), got[-3:], want[-3:])
    for n, (a, b) in enumerate(zip(got, want)):
        ta, tb = a.split(), b.split()
        assert len(ta) == len(tb), (n, a, b)
        for x, y in zip(ta, tb):
            try:
                fx, fy = float(x), float(y)
            except ValueError:
                assert x == y, (n, a, b)
                continue
            assert fx == fy or abs(fx - fy) <= 1.1e-2 * max(abs(fx), abs(fy)), (n, a, b)


@pytest.mark.parametrize("problem,fd", CONSOLE_CASES)
def test_dropin_console_output_is_the_references(built, problem, fd):
    """iLQG() of the product prints what the reference prints with its default console switches (iLQG.c:24-33 and
    :269-374: failed sweeps, the iteration line, the rejected line, the four exits; line_search.c:48-66) — compared with
    tests/golden/trace_*.txt, the output of the reference's own sources (tests/golden/make_goldens.py console)"""
    hip = os.path.join(os.path.dirname(lib_path("oracle")), "libdrv_%s_fd%d_hip.so" % (problem, fd))
    with open(os.path.join(os.path.dirname(__file__), "golden", "trace_%s_fd%d.txt" % (problem, fd))) as f:
        want = f.read()
    assert want.count("iter:") >= 10
    _same_console(console_of(hip, problem, fd), want)
    # debug_level 0 silences everything the reference puts under a threshold
    quiet = console_of(hip, problem, fd, debug_level=0)
    assert [ln for ln in quiet.splitlines() if ln and not ln.startswith("non-positive")] == [want.splitlines()[-1]], quiet
