"""Where the reference build exists (this container: compiled from
/root/reference; GPU box: the prebuilt oracle/_ref/*.so shipped with the
snapshot), check the CPU restatement against it directly on fresh seeded
inputs — exact equality, beyond the committed fixtures."""
import numpy as np
import pytest

from conftest import load_package, ref_available
from oracle.harness import CAR_PARAMS, CONSOLE_CASES, Driver, Kernels, console_of, lib_path

pytestmark = pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no reference sources here)")


@pytest.mark.parametrize("fd", [0, 1])
def test_randomised_solves_match_reference(oracle_built, fd):
    synth = load_package().synth
    x0s, u0s = synth.car_batch(3, first=5000)
    for b in range(3):
        out = []
        for kind in ("ref", "oracle"):
            d = Driver(lib_path(kind, full_ddp=fd), 500, CAR_PARAMS, dict(max_iter=60))
            assert d.init(x0s[b], u0s[b]) == 1
            rc = d.solve()
            out.append((rc, d.scalars(), d.traj(0), d.gains(), d.trace()))
            d.close()
        a, o = out
        assert a[0] == o[0] and a[1] == o[1]
        for i in (2, 3):
            assert np.array_equal(a[i][0], o[i][0]) and np.array_equal(a[i][1], o[i][1])
        for k in a[4]:
            assert np.array_equal(a[4][k], o[4][k]), k


def test_random_boxqp_match_reference(oracle_built):
    Kr, Ko = Kernels(lib_path("ref")), Kernels(lib_path("oracle"))
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(1, 9))
        A = rng.standard_normal((n, n))
        M = A @ A.T + 10.0 ** rng.uniform(-6, 0) * np.eye(n)  # positive definite: rc -2 (which prints) is avoided
        H = np.array([M[r, c] for c in range(n) for r in range(c + 1)])
        g = rng.standard_normal(n)
        lo = -np.abs(rng.standard_normal(n)); hi = np.abs(rng.standard_normal(n))
        x0 = rng.standard_normal(n)
        a, o = Kr.boxqp(H, g, lo, hi, x0), Ko.boxqp(H, g, lo, hi, x0)
        assert a["rc"] == o["rc"] and a["n_free"] == o["n_free"]
        assert np.array_equal(a["x"], o["x"]) and np.array_equal(a["clamp"], o["clamp"])


@pytest.mark.skipif(not __import__("os").path.exists(lib_path("pure")), reason="oracle/_ref/libref_pure.so not built")
def test_dense_helpers_match_the_reference_built_without_any_stand_in(oracle_built):
    """cholesky.c, matMult.c and printMat.c are the reference files that include neither mex.h nor a generated header:
    _ref/libref_pure.so is those three compiled with nothing but -I/root/reference -DPRNT=printf.  The restatement's
    cholesky_tri / cholesky_tri_inv / addMulVec / addSquareTri / addMul2Tri equal it exactly on fresh inputs."""
    Kp, Ko = Kernels(lib_path("pure")), Kernels(lib_path("oracle"))
    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(1, 17))
        A = rng.standard_normal((n, n))
        M = A @ A.T + 10.0 ** rng.uniform(-8, 0) * np.eye(n) if trial % 3 else (A + A.T) / 2
        P = np.array([M[r, c] for c in range(n) for r in range(c + 1)])
        okp, Up = Kp.cholesky(P, n)
        oko, Uo = Ko.cholesky(P, n)
        assert okp == oko
        if okp:
            assert np.array_equal(Up, Uo) and np.array_equal(Kp.cholesky_inv(Up, n), Ko.cholesky_inv(Uo, n))
        m = int(rng.integers(1, n + 1))
        V = np.array([(A @ A.T)[r, c] for c in range(n) for r in range(c + 1)])
        fx, fu, vx = rng.standard_normal(n * n), rng.standard_normal(n * m), rng.standard_normal(n)
        assert np.array_equal(Kp.add_mul_vec(rng.standard_normal(m) * 0 + 1.5, vx, fu, n, m), Ko.add_mul_vec(np.full(m, 1.5), vx, fu, n, m))
        b_uu, b_xx, b_xu = rng.standard_normal(m * (m + 1) // 2), rng.standard_normal(n * (n + 1) // 2), rng.standard_normal(n * m)
        assert np.array_equal(Kp.add_square_tri(b_uu, V, fu, n, m), Ko.add_square_tri(b_uu, V, fu, n, m))
        assert np.array_equal(Kp.add_square_tri(b_xx, V, fx, n, n), Ko.add_square_tri(b_xx, V, fx, n, n))
        assert np.array_equal(Kp.add_mul2_tri(b_xu, V, fx, n, n, fu, n, m), Ko.add_mul2_tri(b_xu, V, fx, n, n, fu, n, m))



@pytest.mark.parametrize("problem,fd", CONSOLE_CASES)
def test_console_fixture_is_what_the_reference_prints(oracle_built, problem, fd):
    """tests/golden/trace_*.txt (the iteration lines the product's iLQG() is compared with on the GPU,
    tests/test_gpu_dropin.py) are the reference's own console output with its default switches"""
    import os
    lib = os.path.join(os.path.dirname(lib_path("ref")), "libref_%s_fd%d_trace.so" % (problem, fd))
    if not os.path.exists(lib):
        pytest.skip("trace build of the reference not present")
    with open(os.path.join(os.path.dirname(__file__), "golden", "trace_%s_fd%d.txt" % (problem, fd))) as f:
        assert console_of(lib, problem, fd) == f.read()
