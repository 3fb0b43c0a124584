"""Build container only: what the GENERATED CarParking pair computes in calc_derivs() — the records the reference's own
back_pass.c consumes in oracle/_ref, and every kernel's parity target — against derivatives taken here, independently, from
the REFERENCE's definition of the problem (examples/CarParking/optDefCar.mac, parsed as text by the Maxima-subset reader of
tests/test_defs_match_reference.py; testCar.m's parameter values): f, L and F differentiated by sympy with the auxiliary s
substituted, evaluated in 40-digit arithmetic at the states and inputs of a roll-out, laid out as iLQG_func.tem:262-291 /
matMult.h:4-9 prescribe (fx column-major, packed upper triangles, fxx[j + i*sizeofQxx] ...).  This checks the one stand-in the
oracle's pins still lean on besides mex.h — the expression printer with its auxiliary reuse (tools/gen_problem.py in place
of Maxima's gentran) — against a second derivation that shares nothing with it but sympy.diff.  Skips where
/root/reference is absent (the GPU box); stores nothing of the .mac text."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")


def tri(n):
    return n * (n + 1) // 2


@pytest.mark.parametrize("kind", ["oracle", "ref"])
def test_carparking_records_equal_derivatives_of_the_reference_definition(kind):
    import sympy as sp
    import mpmath as mp
    import subprocess
    from test_defs_match_reference import read_mac, to_sympy
    from oracle.harness import CAR_PARAMS, CAR_X0, Driver, lib_path

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"])
    M = read_mac(os.path.join(REF, "CarParking", "optDefCar.mac"))
    table = {}

    def sym(name):
        return table.setdefault(name, sp.Symbol(name, real=True))

    aux = {sym(n): to_sympy(t, sym, M["functions"]) for n, t in M["aux"].items()}
    x = [sym(n) for n in M["x"]]
    u = [sym(n) for n in M["u"]]
    f = [to_sympy(M["f"][n], sym, M["functions"]).subs(aux) for n in M["x"]]
    L = to_sympy(M["L"], sym, M["functions"]).subs(aux)
    F = to_sympy(M["F"], sym, M["functions"]).subs(aux)
    values = {sym("%s[%d]" % (k, i)) if len(v) > 1 else sym(k): mp.mpf(vi) for k, v in CAR_PARAMS.items() for i, vi in enumerate(v)}
    n, m = len(x), len(u)
    sxx, suu, nm = tri(n), tri(m), n * m

    # the entries of a step's record in the host order (harness.derivs): cx cxx cu cuu cxu fx fu lower upper fxx fuu fxu ...
    exprs = []
    exprs += [sp.diff(L, x[i]) for i in range(n)]
    exprs += [sp.diff(L, x[r], x[c]) for c in range(n) for r in range(c + 1)]                      # UTRI_MAT_IDX(r, c) = c(c+1)/2 + r
    exprs += [sp.diff(L, u[i]) for i in range(m)]
    exprs += [sp.diff(L, u[r], u[c]) for c in range(m) for r in range(c + 1)]
    exprs += [sp.diff(L, x[r], u[c]) for c in range(m) for r in range(n)]                          # cxu[r + c n]
    exprs += [sp.diff(f[r], x[c]) for c in range(n) for r in range(n)]                             # fx[r + c n]
    exprs += [sp.diff(f[r], u[c]) for c in range(m) for r in range(n)]                             # fu[r + c n]
    first = len(exprs)
    tens = []
    tens += [sp.diff(f[i], x[r], x[c]) for i in range(n) for c in range(n) for r in range(c + 1)]  # fxx[j + i sxx]
    tens += [sp.diff(f[i], u[r], u[c]) for i in range(n) for c in range(m) for r in range(c + 1)]  # fuu[j + i suu]
    tens += [sp.diff(f[i], x[r], u[c]) for i in range(n) for c in range(m) for r in range(n)]      # fxu[r + c n + i nm]
    fin = [sp.diff(F, x[i]) for i in range(n)] + [sp.diff(F, x[r], x[c]) for c in range(n) for r in range(c + 1)]
    args = x + u + list(values)
    ev = sp.lambdify(args, exprs + tens, "mpmath")
    ev_fin = sp.lambdify(x + list(values), fin, "mpmath")

    N = 40
    rng = np.random.default_rng(7)
    u0 = 0.3 * rng.standard_normal((N, m))
    d = Driver(lib_path(kind, "carparking", 1), N, CAR_PARAMS, dict(max_iter=1))
    assert d.init(np.array(CAR_X0) + 0.1, u0) == 1
    d.calc_derivs()
    rec, rfin = d.derivs()
    xs, us = d.traj(0)
    d.close()
    mp.mp.dps = 40
    worst = 0.0
    off_t = first + 2 * m  # (lower, upper between the first-order part and the tensors)
    for k in range(0, N, 3):
        pt = [mp.mpf(float(v)) for v in xs[k]] + [mp.mpf(float(v)) for v in us[k]] + list(values.values())
        want = [float(v) for v in ev(*pt)]
        got = list(rec[k][:first]) + list(rec[k][off_t:off_t + n * (sxx + suu + nm)])
        assert len(got) == len(want)
        for g, w in zip(got, want):
            worst = max(worst, abs(g - w) / max(1.0, abs(w)))
    want = [float(v) for v in ev_fin(*([mp.mpf(float(v)) for v in xs[N]] + list(values.values())))]
    for g, w in zip(rfin, want):
        worst = max(worst, abs(g - w) / max(1.0, abs(w)))
    assert worst < 1e-12, worst  # (measured: 2.8e-16, oracle and reference build alike)
