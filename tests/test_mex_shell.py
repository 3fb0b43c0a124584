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
