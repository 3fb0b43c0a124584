"""Problem pairs that carry the reference templates' LITERAL C (tools/fill_reference_template.py: every line of
iLQG_func.tem / iLQG_problem.tem outside <<...>> kept, the blocks filled by tools/gen_problem.py's printers) through every
consumer of a problem file: gcc + the reference's own solver sources, the oracle, and the gfx950 device wrapper of
ilqg_kernels.hip.  Answers "does a Maxima-generated file link unchanged" for the part of such a file that is known here:
the template's own text (limitsU with index arrays, pointer walks and a switch; forward_pass, calc_derivs, init_opt,
update_multipliers_*; the aux_ / daux_ / mu_ macros that stay defined to the end of the translation unit).

The pairs are made and built by __graft_entry__.build() in the build container only (they are derivatives of the
reference's files: oracle/_ref/tem/ is neither tracked nor shipped); the libraries travel with oracle/_ref.  Everything
here skips where they are absent."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference"
TEM = os.path.join(ROOT, "oracle", "_ref", "tem")
HIP = os.path.join(ROOT, "oracle", "_ref", "hip")
PROBLEMS = ("carparking", "hxtest", "almix")


def tem_lib(kind, problem, fd):
    name = {"ref": "libref_%s_tem_fd%d.so", "oracle": "liboracle_%s_tem_fd%d.so"}[kind] % (problem, fd)
    return os.path.join(ROOT, "oracle", "_ref", name)


def need(path):
    if not os.path.exists(path):
        pytest.skip("%s absent (made in the build container only)" % os.path.relpath(path, ROOT))
    return path


@pytest.fixture(scope="module")
def built():
    if os.path.exists(os.path.join(REF, "iLQG_func.tem")):
        import __graft_entry__ as g
        g.build_for_tests()
    return True


@pytest.mark.parametrize("problem", PROBLEMS)
def test_every_literal_line_of_the_templates_is_kept(built, problem):
    if not os.path.exists(os.path.join(REF, "iLQG_func.tem")):
        pytest.skip("reference not present")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fill_reference_template as F
    for which, name in (("problem", "iLQG_problem.h"), ("func", "iLQG_func.c")):
        tem = open(os.path.join(REF, "iLQG_%s.tem" % which)).read()
        out = open(need(os.path.join(TEM, problem + "_tem", name))).read()
        assert len(F.literal_lines(tem)) > (30 if which == "problem" else 250)
        assert F.missing_literals(tem, out) == []
        assert "<<" not in out and ">>" not in out
    func = open(os.path.join(TEM, problem + "_tem", "iLQG_func.c")).read()
    # the text in question is there as the template has it
    for piece in ("int lower_idx[N_U], upper_idx[N_U], *idx_;", "for(i= 0; i<N_U; i++, hx_+= N_X, h_sign++) {", "switch(idx_[i]) {",
                  "#define mcond(cond, a, dummy, b) ((cond)? a: b)", "for(k= N-1; k>=0; k--, t--, m--) {", "if(init) return 1;"):
        assert piece in func, piece
    assert "ILQG_" not in func  # none of this library's additive hints


@pytest.mark.parametrize("fd", [0, 1])
def test_reference_solver_and_oracle_on_the_template_pair_equal_the_goldens(built, fd):
    """the committed goldens were recorded from the reference's solver over tools/gen_problem.py's own pair; the
    reference's solver (and the oracle) over the template-literal pair reproduce them bit for bit"""
    from test_oracle_golden import check_almix, check_car_single, check_hxtest
    for kind in ("ref", "oracle"):
        check_car_single(need(tem_lib(kind, "carparking", fd)), fd)
        check_hxtest(need(tem_lib(kind, "hxtest", fd)), fd)
        check_almix(need(tem_lib(kind, "almix", fd)), fd)


def test_device_wrapper_keeps_the_template_pair_in_registers(built):
    """k_derivs, the fused backward kernel and the search kernel of the template-literal pair: no scratch memory, and
    the registers of the hint-free pair emitted by tools/gen_problem.py --plain (same expressions, other C around them)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as K
    want = re.compile(r"^(k_derivs|k_backward<2>|k_search<0, true>|k_rollout<[012]>)$")
    for fd in (0, 1):
        tem = {k["name"]: k for k in K.kernels(need(os.path.join(HIP, "libilqg_carparking_tem_fd%d_hip.so" % fd))) if want.match(k["name"])}
        plain = {k["name"]: k for k in K.kernels(os.path.join(ROOT, "ddp-generator_amd", "lib", "libilqg_carparking_plain_fd%d_hip.so" % fd)) if want.match(k["name"])}
        assert set(tem) == set(plain) and len(tem) == 6
        for name in tem:
            # (FULL_DDP = 1 in the lane mapping: the backward kernel holds 139 record entries per lane and spills in EVERY
            # build of CarParking, the shipped one included — the template's text adds nothing to it)
            assert tem[name]["scratch"] == plain[name]["scratch"] and (fd == 1 or tem[name]["scratch"] == 0), (name, tem[name])
            assert abs(tem[name]["vgpr"] - plain[name]["vgpr"]) <= 8, (name, tem[name]["vgpr"], plain[name]["vgpr"])
    hx = {k["name"]: k for k in K.kernels(need(os.path.join(HIP, "libilqg_hxtest_tem_fd0_hip.so")))}
    assert hx["k_derivs"]["scratch"] == 0 and hx["k_backward<2>"]["scratch"] == 0


@pytest.mark.parametrize("problem", PROBLEMS)
def test_no_macro_of_the_problem_file_reaches_the_kernels(built, problem):
    """the #undef list the wrapper includes behind the file names every macro the file defines; and no identifier of the
    hand-written device code coincides with one (it would have been rewritten before the list existed)"""
    func = open(need(os.path.join(TEM, problem + "_tem", "iLQG_func.c"))).read()
    defined = set(re.findall(r"^[ \t]*#[ \t]*define[ \t]+([A-Za-z_]\w*)", func, flags=re.M))
    assert {"mcond", "sec", "csc"} <= defined and any(d.startswith("aux_") for d in defined)
    undefs = open(os.path.join(ROOT, "ddp-generator_amd", "build_tem", problem + "_tem_fd0", "ilqg_problem_undefs.h")).read()
    assert set(re.findall(r"^#undef (\w+)$", undefs, flags=re.M)) == defined
    csrc = os.path.join(ROOT, "ddp-generator_amd", "csrc")
    text = open(os.path.join(csrc, "ilqg_kernels.hip")).read()
    text = text[text.index('#include "ilqg_problem_undefs.h"'):]
    for h in sorted(os.listdir(csrc)):
        if h.endswith((".hpp", ".inc")) or h == "ilqg_shim.h":
            text += open(os.path.join(csrc, h)).read()
    text = re.sub(r"//[^\n]*|/\*.*?\*/", " ", text, flags=re.S)
    used = set(re.findall(r"[A-Za-z_]\w*", text))
    assert not (used & defined), sorted(used & defined)
    assert not [u for u in used if re.match(r"(aux_|daux_|mu_[fl][ei]_)", u)]


@pytest.mark.gpu
@pytest.mark.parametrize("fd", [0, 1])
def test_template_pair_on_the_device_golden(fd):
    """the reference-build goldens through the device libraries built from the template-literal pairs"""
    need(os.path.join(HIP, "libilqg_carparking_tem_fd%d_hip.so" % fd))
    from conftest import load_package
    load_package()
    from ddp_generator_amd import ilqg
    ilqg.add_library_dir(HIP)
    from test_gpu_multipliers import check_almix_strict
    from test_gpu_parity import check_hxtest_golden, check_single_pass
    check_single_pass(ilqg, "carparking_tem", fd)
    check_hxtest_golden(ilqg, "hxtest_tem", fd)
    check_almix_strict(ilqg, "almix_tem", fd)
