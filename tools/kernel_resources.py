#!/usr/bin/env python3
"""Registers, scratch and LDS of the kernels inside a built object or library (no GPU needed).

    python tools/kernel_resources.py [--only REGEX] file.o|file.so [...]

Takes the .hip_fatbin section out of the file (objcopy), unbundles the gfx950 code object (clang-offload-bundler) and
reads the kernel descriptors' metadata (llvm-readelf --notes): VGPRs, SGPRs, spills, scratch bytes per lane
(private_segment_fixed_size) and static LDS bytes (group_segment_fixed_size) per kernel, names demangled.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_object(path, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
    co = os.path.join(tmp, "k.co")
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET,
                           "--input=" + fat, "--output=" + co])
    return co


def kernels(path):
    """[{name, vgpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] of the gfx950 kernels in an object or shared library"""
    with tempfile.TemporaryDirectory() as tmp:
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", code_object(path, tmp)], text=True)
    out, cur = [], {}
    keys = {".name": "name", ".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill",
            ".sgpr_spill_count": "sgpr_spill", ".private_segment_fixed_size": "scratch", ".group_segment_fixed_size": "lds"}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s+(\S+)", line)
        if not m or m.group(1) not in keys:
            continue
        k = keys[m.group(1)]
        if k in cur:  # the next kernel's block begins
            out.append(cur)
            cur = {}
        cur[k] = m.group(2) if k == "name" else int(m.group(2))
    if cur:
        out.append(cur)
    out = [k for k in out if "name" in k and "vgpr" in k]
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(out, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        k["name"] = re.sub(r"^void ", "", re.sub(r"\(.*$", "", n))
    return out


def table(path, only=None):
    rows = [k for k in kernels(path) if not only or re.search(only, k["name"])]
    lines = ["%-44s %5s %5s %7s %8s %7s" % ("kernel", "VGPR", "SGPR", "spills", "scratch", "LDS")]
    for k in rows:
        lines.append("%-44s %5d %5d %7d %8d %7d" % (k["name"][:44], k["vgpr"], k["sgpr"], k.get("vgpr_spill", 0) + k.get("sgpr_spill", 0),
                                                     k.get("scratch", 0), k.get("lds", 0)))
    return "\n".join(lines)


if __name__ == "__main__":
    args = sys.argv[1:]
    only = None
    if "--only" in args:
        i = args.index("--only")
        only = args[i + 1]
        del args[i:i + 2]
    if not args:
        print(__doc__)
        sys.exit(2)
    for p in args:
        print("== %s" % p)
        print(table(p, only))
