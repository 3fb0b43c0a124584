"""The build path INTEGRATION.md prints for make_iLQG.m (reference make_iLQG.m:43-52, 74-86: Maxima writes the generated
pair into <problem>_gen_files/, the build then names that directory): a pair that lives OUTSIDE problems/, under a title
of its own, built by `make PROBLEMS=<title> PROBLEM_DIR=<dir>`.  __graft_entry__.build() runs exactly that command on a
copy of the hint-free CarParking pair (title `parkdemo`, libraries in ddp-generator_amd/lib_oot/)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden

OOT_LIBDIR = os.path.join(ROOT, "ddp-generator_amd", "lib_oot")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build_for_tests()
    from ddp_generator_amd import ilqg
    ilqg.add_library_dir(OOT_LIBDIR)
    return ilqg


def test_documented_make_line_builds_the_library(built, tmp_path):
    """the literal command of INTEGRATION.md section 1 on a fresh directory and a fresh title: library, exports, facts"""
    import __graft_entry__ as g
    gen = tmp_path / "other_gen_files"
    gen.mkdir()
    src = os.path.join(ROOT, "ddp-generator_amd", "build_oot", g.OOT_TITLE + "_gen_files")
    for f in ("iLQG_problem.h", "iLQG_func.c"):
        (gen / f).write_text(open(os.path.join(src, f)).read())
    line = ["make", "-s", "-C", os.path.join(ROOT, "ddp-generator_amd", "csrc"), "PROBLEMS=othertitle", "PROBLEM_DIR=%s" % gen,
            "LIBDIR=%s" % (tmp_path / "lib"), "OBJDIR=%s" % (tmp_path / "obj"), "FDS=0", "STRICT=0"]
    subprocess.check_call(line, stderr=subprocess.DEVNULL)
    lib = C.CDLL(str(tmp_path / "lib" / "libilqg_othertitle_fd0_hip.so"))
    for name in ("iLQG", "back_pass", "line_search", "boxQP", "forward_pass", "calc_derivs", "init_opt", "ilqg_batch_create",
                 "ilqg_solve_single", "ilqg_multi_create"):
        assert hasattr(lib, name), name
    dims = (C.c_int * 8)()
    lib.ilqg_problem_dims(dims)
    assert (dims[0], dims[1], dims[2]) == (4, 2, 0)
    # two titles at once need the per-title form
    bad = subprocess.run(["make", "-s", "-n", "-C", os.path.join(ROOT, "ddp-generator_amd", "csrc"), "PROBLEMS=a b", "PROBLEM_DIR=%s" % gen],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "PROBLEM_SRC_<title>" in bad.stderr


def test_out_of_tree_library_has_the_problem(built):
    for fd in (0, 1):
        p = built.Problem("parkdemo", fd)
        assert (p.nx, p.nu, p.full_ddp) == (4, 2, fd)
        assert dict(p.params) == dict(cf=4, cu=2, cx=2, d=1, h=1, limA=2, limW=2, pf=4, px=2)
    assert os.path.dirname(built.library_path("parkdemo", 0)) == OOT_LIBDIR


@pytest.mark.gpu
@pytest.mark.parametrize("fd", [0, 1])
def test_out_of_tree_library_golden(fd):
    """the reference-build goldens of CarParking through the library built from the out-of-tree directory"""
    from conftest import load_package
    load_package()
    from ddp_generator_amd import ilqg
    ilqg.add_library_dir(OOT_LIBDIR)
    from test_gpu_parity import check_single_pass
    check_single_pass(ilqg, "parkdemo", fd)
