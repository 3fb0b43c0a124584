"""Boundary check (build container only): the reference's MEX shell iLQG_mex.c (iLQG_mex.c:19-144), read where it
lies under /root/reference, compiles UNCHANGED against the headers this library ships in include/ and a generated
problem header — every type, field, macro and prototype it uses is there.  The MEX API itself is declared by
tests/mex_api/mex.h (declarations only).  Skipped where the reference is absent (GPU box)."""
import os
import subprocess

import pytest

from conftest import ROOT

MEX_SHELL = "/root/reference/iLQG_mex.c"


@pytest.mark.skipif(not os.path.exists(MEX_SHELL), reason="reference sources not present")
@pytest.mark.parametrize("host", ["-DHAVE_OCTAVE", "-DMATLAB_MEX_FILE"])  # mkoctfile --mex / mex (make_iLQG.m:61-86)
@pytest.mark.parametrize("problem,fd", [("carparking", 0), ("carparking", 1), ("brachi", 0), ("synth16x8", 1)])
def test_reference_mex_shell_compiles_against_shipped_headers(problem, fd, host):
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration",
           "-Werror=incompatible-pointer-types", "-Werror=int-conversion", "-DFULL_DDP=%d" % fd, "-DPRNT=mexPrintf", host,
           "-I", os.path.join(ROOT, "tests", "mex_api"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "problems", problem), "-x", "c", "-"]
    with open(MEX_SHELL, "rb") as f:
        r = subprocess.run(cmd, stdin=f, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


SHELLS = ["ilqg_mex.c", "ilqg_batch_mex.c"]


@pytest.mark.parametrize("host", ["-DHAVE_OCTAVE", "-DMATLAB_MEX_FILE"])
@pytest.mark.parametrize("shell", SHELLS)
def test_shipped_mex_shells_compile(shell, host):
    """the MEX shells this repository ships (ddp-generator_amd/host/: the thin single-trajectory shell replacing
    iLQG_mex.c:19-144 and the batch shell) against the declaration-only MEX API header and include/ilqg_batch.h;
    every library function they call is declared there and exported by the library (tests/test_cabi.py)"""
    src = os.path.join(ROOT, "ddp-generator_amd", "host", shell)
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", host,
           "-I", os.path.join(ROOT, "tests", "mex_api"), "-I", os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_mex_make_target_is_guarded():
    """`make mex` builds the shells with mkoctfile / mex when one is on PATH and says so when none is (this image)"""
    import shutil
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "ddp-generator_amd", "csrc"), "mex", "PROBLEMS=carparking"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if not (shutil.which("mkoctfile") or shutil.which("mex")):
        assert "neither mkoctfile nor mex" in r.stdout + r.stderr
