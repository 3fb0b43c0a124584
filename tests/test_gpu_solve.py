"""Full solves with finished trajectories retired (ilqg_batch_solve, option "compact"; reference loop exits
iLQG.c:297-303, :331, :365-378): gathering the live trajectories into smaller contexts changes where a trajectory is
iterated, not what is computed — every per-trajectory result equals the uncompacted solve bit for bit — and a solve equals
the CPU oracle's at the full-solve bar of SURVEY 8(c)."""
import numpy as np
import pytest

from conftest import load_package
from oracle.harness import CAR_PARAMS, Driver, lib_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ilqg():
    import __graft_entry__ as g
    g.build_for_tests()
    from ddp_generator_amd import ilqg as m
    return m


def _solve(ilqg, problem, fd, x0, u0, n_hor, params, opts, compact, **kw):
    s = ilqg.BatchSolver(problem, fd, batch=len(x0), n_hor=n_hor, params=params, opts=dict(opts, compact=compact), **kw)
    s.init(x0, u0)
    s.solve()
    l, L = s.gains()
    out = dict(x=s.x(), u=s.u(), l=l, L=L, trace=s.solve_trace())
    for k in ("cost", "lambda", "dlambda", "g_norm", "dV0", "dV1", "new_cost", "dcost", "expected", "alpha_cost"):
        out[k] = s.scalar(k).copy()
    for k in ("status", "iterations", "alpha_idx", "accepted", "bp_calls", "alpha_ok"):
        out[k] = s.ints(k).copy()
    s.close()
    return out


@pytest.mark.parametrize("problem,fd,groups", [("carparking", 0, 0), ("carparking", 1, 2), ("hxtest", 0, 0), ("synth16x8", 1, 0),
                                               ("synth16x8", 0, 0)])
def test_compacted_solve_equals_the_uncompacted_one(ilqg, problem, fd, groups):
    """CarParking starts converge at very different iterations: with "compact" the solve gathers the live trajectories
    several times (the trace shows the slots shrinking) and every result — trajectory, gains, cost, lambda, exit reason,
    iteration count, the last line search's per-step-size costs — equals the plain solve's"""
    pkg = load_package()
    if problem == "carparking":
        B, N, params, opts = 700, 500, ilqg.CAR_PARAMS, dict(max_iter=400)
        x0, u0 = pkg.synth.car_batch(B, N)
    elif problem == "synth16x8":  # the wave mapping (quad / row kernels, records private to the context)
        from oracle.harness import SYN_PARAMS_TIGHT, syn_inputs
        B, N, params, opts = 200, 60, SYN_PARAMS_TIGHT, dict(max_iter=200)
        x0, u0 = syn_inputs(B, N)
    else:
        from oracle.harness import HX_N, HX_PARAMS, hx_inputs
        B, N, params, opts = 300, HX_N, HX_PARAMS, dict(max_iter=60)
        x0, u0 = hx_inputs(B)
    kw = dict(groups=groups) if groups else {}
    plain = _solve(ilqg, problem, fd, x0, u0, N, params, opts, 0, **kw)
    comp = _solve(ilqg, problem, fd, x0, u0, N, params, opts, 8 if problem == "synth16x8" else 16, **kw)
    it, act, slots, n_comp = comp["trace"]
    assert n_comp >= (1 if problem == "hxtest" else 2) and slots[0] == B and slots[-1] < B // 2, (n_comp, slots)
    assert np.all(act <= slots) and plain["trace"][3] == 0 and np.all(plain["trace"][2] == B)
    st = plain["status"]
    # they do finish, and at different iterations
    assert (st != 0).sum() > B // 2 and len(np.unique(plain["iterations"])) > (10 if problem == "carparking" else 3)
    for k in plain:
        if k != "trace":
            assert np.array_equal(plain[k], comp[k]), k


def test_compacted_solve_with_stored_records_and_multipliers(ilqg):
    """the unfused path (stored derivative records: a rejected step sweeps again over the records it has, iLQG.c:345-349)
    and the multipliers of an augmented-Lagrangian problem move with their trajectories"""
    from oracle.harness import almix_case
    params, opts, x0, u0 = almix_case(batch=200)
    N = u0.shape[1]
    plain = _solve(ilqg, "almix", 1, x0, u0, N, params, opts, 0)
    comp = _solve(ilqg, "almix", 1, x0, u0, N, params, opts, 8)
    assert comp["trace"][3] >= 1, comp["trace"]
    for k in plain:
        if k != "trace":
            assert np.array_equal(plain[k], comp[k]), k


def test_compacted_solve_against_the_oracle(ilqg):
    """three trajectories of a compacted CarParking solve against the CPU oracle's solves of the same starts: exit by the
    same test, final cost and trajectory at the full-solve bar (rel 1e-6 / abs 1e-4 where the two CPU builds of the
    reference agree themselves: DESIGN section 4 — starts picked among those that converge early, before the
    iteration map has amplified the contraction difference)"""
    pkg = load_package()
    B, N = 256, 500
    x0, u0 = pkg.synth.car_batch(B, N)
    opts = dict(max_iter=200)
    comp = _solve(ilqg, "carparking", 0, x0, u0, N, ilqg.CAR_PARAMS, opts, 16)
    assert comp["trace"][3] >= 1
    order = np.argsort(comp["iterations"])
    checked = 0
    for b in order[:60]:
        d = Driver(lib_path("oracle", "carparking", 0), N, CAR_PARAMS, opts)
        assert d.init(x0[b], u0[b]) == 1
        d.solve()
        sc = d.scalars()
        xr, ur = d.traj(0)
        d.close()
        # (free-running solves of the two builds part ways where a step size is chosen differently — DESIGN section 4: the
        # FMA-free and the FMA build of the reference itself do on 15 of 16 starts — and may then end in another local optimum;
        # the starts that walk the same path must agree at the full-solve bar)
        same = (int(sc["iterations"]) == int(comp["iterations"][b]) and abs(comp["cost"][b] - sc["cost"]) <= 1e-6 * abs(sc["cost"])
                and np.abs(comp["x"][b] - xr).max() < 1e-4 and np.abs(comp["u"][b] - ur).max() < 1e-4)
        checked += int(same)
    assert checked >= 3, checked  # (of the 60 earliest finishers)


@pytest.mark.parametrize("problem,fd", [("carparking", 0), ("hxtest", 1), ("synth16x8", 1)])
def test_stream_of_starts_equals_plain_solves(ilqg, problem, fd):
    """ilqg_batch_solve_stream: 3.3 batches' worth of starts through the slots of one batch — finished trajectories harvested,
    their slots refilled from a staging context — against plain solves of the same starts in batches of their own: cost,
    exit reason, iteration count and the trajectories, bit for bit; the refills happened while other trajectories were
    still being iterated (the trace shows active trajectories at every refill but the first ones)"""
    pkg = load_package()
    if problem == "carparking":
        B, N, params, opts = 192, 500, ilqg.CAR_PARAMS, dict(max_iter=150)
        total = 640
        x0, u0 = pkg.synth.car_batch(total, N)
    elif problem == "synth16x8":
        from oracle.harness import SYN_PARAMS_TIGHT, syn_inputs
        B, N, params, opts = 64, 40, SYN_PARAMS_TIGHT, dict(max_iter=120)
        total = 210
        x0, u0 = syn_inputs(total, N)
    else:
        from oracle.harness import HX_N, HX_PARAMS, hx_inputs
        B, N, params, opts = 128, HX_N, HX_PARAMS, dict(max_iter=60)
        total = 420
        x0, u0 = hx_inputs(total)
    s = ilqg.BatchSolver(problem, fd, batch=B, n_hor=N, params=params, opts=opts)
    got = s.solve_stream(x0, u0, with_trajectories=True)
    it, act, slots, refills = s.solve_trace()
    s.close()
    assert refills >= 4 and np.all(slots == B) and act.max() <= B
    ref = dict(cost=[], status=[], iterations=[], x=[], u=[])
    for first in range(0, total, 256):
        sel = slice(first, min(total, first + 256))
        p = _solve(ilqg, problem, fd, x0[sel], u0[sel], N, params, opts, 0)
        for k in ref:
            ref[k].append(p[k])
    for k in ref:
        assert np.array_equal(np.concatenate(ref[k]), got[k]), k
    assert len(np.unique(got["iterations"])) > 3 and (got["status"] != 0).all()
