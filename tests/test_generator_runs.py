"""The runs of record entries the generated function file writes through ILQG_REC / ILQG_REC_DONE (tools/gen_problem.py
_record_runs; additive): every time-varying entry of the header's list is assigned exactly once —
in a piece or directly —, a piece is closed behind the last mention of its entries and before the ring of 64 slots a
back-end keeps them in is written again, and pieces end where the place in the element is a multiple of 64."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _offsets(n, m):
    off, out = 0, {}
    for name, size in (("x", n), ("u", m), ("lower", m), ("upper", m), ("lower_sign", m), ("upper_sign", m), ("lower_hx", n * m),
                       ("upper_hx", n * m), ("l", m), ("L", m * n), ("c", 1), ("cx", n), ("cxx", n * (n + 1) // 2), ("cu", m),
                       ("cuu", m * (m + 1) // 2), ("cxu", n * m), ("fx", n * n), ("fu", n * m)):
        out[name] = off
        off += size
    return out


@pytest.mark.parametrize("problem", ["synth16x8"])
def test_runs_of_the_first_derivative_function(problem):
    d = os.path.join(ROOT, "problems", problem)
    head = open(os.path.join(d, "iLQG_problem.h")).read()
    src = open(os.path.join(d, "iLQG_func.c")).read()
    n, m = int(re.search(r"#define N_X (\d+)", head).group(1)), int(re.search(r"#define N_U (\d+)", head).group(1))
    at = _offsets(n, m)
    varying = set(re.findall(r"X\((\w+), (\d+)\)", re.search(r"#define ILQG_TIME_VARYING\(X\) (.*)", head).group(1)))
    varying = {(a, int(b)) for a, b in varying}
    body = src[src.index("static int bp_derivsL_first("):]
    body = body[:body.index("\n}\n")]
    direct = {(a, int(b)) for a, b in re.findall(r"X\((\w+), (\d+)\)", re.search(r"#define ILQG_REC_DIRECT\(X\) (.*)", src).group(1))}
    lines = body.split("\n")
    staged_at, plain_at, last_mention, pieces = {}, {}, {}, []
    for i, ln in enumerate(lines):
        code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', ln)
        for a, b in re.findall(r"ILQG_REC\((\w+), (\d+)\)", code):
            last_mention[(a, int(b))] = i
        mo = re.match(r"\s*ILQG_REC\((\w+), (\d+)\)= ", code)
        if mo:
            assert (mo.group(1), int(mo.group(2))) not in staged_at
            staged_at[(mo.group(1), int(mo.group(2)))] = i
        mo = re.match(r"\s*t->(\w+)\[(\d+)\]= ", code)
        if mo:
            plain_at[(mo.group(1), int(mo.group(2)))] = i
        mo = re.match(r"\s*ILQG_REC_DONE\((\w+), (\d+), (\d+)\)", code)
        if mo:
            pieces.append((i, mo.group(1), int(mo.group(2)), int(mo.group(3))))
    # every time-varying entry once: staged or direct, and the direct list says which
    assert set(staged_at) | set(plain_at) == varying and not (set(staged_at) & set(plain_at))
    assert set(plain_at) == direct
    assert len(staged_at) > 0.9 * len(varying)  # (the dense members go through the runs)
    place = lambda e: at[e[0]] + e[1]
    by_place = {place(e): e for e in staged_at}
    covered, prev_line = set(), -1
    for line, member, first, count in pieces:
        assert 1 <= count <= 64
        span = [at[member] + first + j for j in range(count)]
        ents = [by_place[p] for p in span]  # (a piece may reach into the next member: by place)
        assert not (set(ents) & covered)
        covered |= set(ents)
        # closed behind the last mention (assignment or guard) of each of its entries, in front of the next piece's first
        assert all(prev_line < staged_at[e] and last_mention[e] < line for e in ents)
        # one window of the ring: ends at a multiple of 64 or is the end of its run; never wraps onto itself
        assert len({p % 64 for p in span}) == count
        assert (span[-1] + 1) % 64 == 0 or (span[-1] + 1) not in by_place or span[0] // 64 == span[-1] // 64
        prev_line = line
    assert covered == set(staged_at)


def test_cpu_build_sees_plain_assignments():
    """without a back-end's definitions the macros are the plain assignment and nothing"""
    src = open(os.path.join(ROOT, "problems", "synth16x8", "iLQG_func.c")).read()
    assert "#define ILQG_REC(member, index) t->member[index]" in src
    assert re.search(r"#define ILQG_REC_DONE\(member, first, count\)\s*/\*", src)
    assert "#define ILQG_BASIS(index) basis[index]" in src


def _tri(n):
    return n * (n + 1) // 2


@pytest.mark.parametrize("problem,fd", [("carparking", 0), ("carparking", 1), ("hxtest", 1), ("brachi", 1), ("almix", 1), ("synth16x8", 0)])
def test_structural_zero_list_against_evaluated_records(problem, fd):
    """ILQG_STRUCTURAL_ZERO / _FULL (additive lists of the generated header: the record entries that are identically 0, which
    the fused backward sweep of the lane mapping leaves out of its products, ilqg_device.hpp NoZeros): every listed entry
    IS 0.0 in records the generated callbacks evaluate along a rolled-out trajectory (the CPU checker's calc_derivs, the
    same function file), the lists and the time-varying lists do not overlap, and for CarParking they are the zeros SURVEY
    Appendix A.4's model has (fx: 7 of 16, fu: 4 of 8, cxx: 8 of 10, all of cxu)"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import harness as H
    head = open(os.path.join(ROOT, "problems", problem, "iLQG_problem.h")).read()
    n, m = int(re.search(r"#define N_X (\d+)", head).group(1)), int(re.search(r"#define N_U (\d+)", head).group(1))

    def entries(macro, full):
        ms = re.findall(r"#define %s\(X\)(.*)" % macro, head)
        txt = ms[0] if (full is None or len(ms) == 1) else ms[0 if full else 1]
        return {(a, int(b)) for a, b in re.findall(r"X\((\w+), (\d+)\)", txt)}
    zeros = entries("ILQG_STRUCTURAL_ZERO", None) | (entries("ILQG_STRUCTURAL_ZERO_FULL", True) if fd else set())
    varying = entries("ILQG_TIME_VARYING", None) | (entries("ILQG_TIME_VARYING_FULL", True) if fd else set())
    assert zeros and not (zeros & varying)
    # the record as oracle/driver.c lays it out (RecLayout of ilqg_device.hpp: the same order)
    off, at = 0, {}
    for name, size in (("cx", n), ("cxx", _tri(n)), ("cu", m), ("cuu", _tri(m)), ("cxu", n * m), ("fx", n * n), ("fu", n * m),
                       ("lower", m), ("upper", m)) + ((("fxx", n * _tri(n)), ("fuu", n * _tri(m)), ("fxu", n * n * m)) if fd else ()):
        at[name] = off
        off += size
    path = H.lib_path("oracle", problem, fd)
    if not os.path.exists(path):
        pytest.skip("oracle library not built")
    rng = np.random.default_rng(5)
    if problem == "carparking":
        N, params = 40, H.CAR_PARAMS
        x0, u0 = np.array([1.0, 1.0, 4.7, 0.0]), 0.3 * rng.standard_normal((N, m))
    elif problem == "hxtest":
        N, params = 30, H.HX_PARAMS
        (x0, u0) = H.hx_inputs(1)
        x0, u0, N = x0[0], u0[0], u0.shape[1]
    elif problem == "synth16x8":
        N, params = 12, H.SYN_PARAMS
        x0, u0 = H.syn_inputs(1, N)
        x0, u0 = x0[0], u0[0]
    elif problem == "brachi":
        params, opts, x0, u0 = H.brachi_case(20)
        N = u0.shape[0]
    else:
        params, opts, x0, u0 = H.almix_case()
        N = u0.shape[0]
    d = H.Driver(path, N, params, {})
    assert d.init(x0, u0) == 1 and d.calc_derivs() == 1
    rec, _ = d.derivs()
    d.close()
    for member, index in sorted(zeros):
        col = rec[:, at[member] + index]
        assert np.all(col == 0.0), (member, index, col[:3])
    if problem == "carparking" and fd == 0:
        count = lambda mem: sum(1 for a, _ in zeros if a == mem)
        assert (count("fx"), count("fu"), count("cxx"), count("cxu")) == (7, 4, 8, 8)
