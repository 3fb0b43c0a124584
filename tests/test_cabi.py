"""CPU-side checks of the product library: it loads without a GPU, exports every
symbol the public headers declare, and refuses to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build_for_tests()
    from ddp_generator_amd import ilqg
    return ilqg


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_ \*]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text, flags=re.M)
    return sorted(set(names))


@pytest.mark.parametrize("fd", [0, 1])
def test_library_exports_declared_symbols(built, fd):
    lib = C.CDLL(built.library_path("carparking", fd))
    wanted = set(declared_functions("ilqg_batch.h"))
    # the reference's link-time solver symbols; the generated side
    # (forward_pass, calc_derivs, init_opt, ...) is exported by the linked problem file
    wanted |= {"iLQG", "standard_parameters", "setOptParam", "makeCandidateNominal", "printParams", "back_pass",
               "line_search", "boxQP", "forward_pass", "calc_derivs", "init_opt", "update_multipliers",
               "ilqg_release", "printVec", "printTri", "printMat", "addMulVec", "addSquareTri", "addMul2Tri",
               "cholesky_tri", "cholesky_tri_inv"}
    assert len(wanted) > 40
    missing = [n for n in sorted(wanted) if not hasattr(lib, n)]
    assert not missing, missing


def test_problem_facts_and_param_table(built):
    p = built.Problem("carparking", 0)
    assert (p.nx, p.nu, p.full_ddp) == (4, 2, 0)
    assert p.rec_dev == 55 and p.rec_host == 75          # 55 + u[2] = 57 doubles read per step
    assert dict(p.params) == dict(cf=4, cu=2, cx=2, d=1, h=1, limA=2, limW=2, pf=4, px=2)
    p1 = built.Problem("carparking", 1)
    assert p1.rec_dev == 55 + 84 and p1.full_ddp == 1    # + fxx, fuu, fxu


def test_no_device_means_loud_failure(built):
    p = built.Problem("carparking", 0)
    if p.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(built.IlqgError):
        built.BatchSolver("carparking", 0, batch=4, n_hor=10)


def test_missing_library_is_an_error(built):
    with pytest.raises(built.IlqgError):
        built.load_library("no_such_problem", 0)
