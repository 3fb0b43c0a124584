"""Static instruction mix of a stretch of device assembly by region (line ranges of the .s file).
   python tools/experiments/isa_mix.py file.s name:first-last ..."""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
def cat(op):
    if not op.startswith("v_"):
        if op.startswith("ds_"): return "lds"
        if op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_"): return "vmem"
        if op == "s_nop": return "s_nop"
        if op == "s_waitcnt": return "waitcnt"
        return "salu"
    if "f64" in op and "dpp" in op and "mov" not in op: return "fp64_dpp"
    if "f64" in op: return "fp64"
    if op.startswith("v_mov_b64_dpp"): return "mov_dpp"
    if "readlane" in op or "readfirstlane" in op: return "readlane"
    if "writelane" in op: return "writelane"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_cmp"): return "cmp"
    return "int"
order = ["fp64_dpp", "fp64", "mov_dpp", "readlane", "writelane", "mov", "cndmask", "cmp", "int", "lds", "vmem", "salu", "s_nop", "waitcnt"]
print("%-28s %6s | " % ("region", "VALU") + " ".join("%9s" % o for o in order))
for spec in sys.argv[2:]:
    name, r = spec.split(":")
    a, b = map(int, r.split("-"))
    c = collections.Counter()
    for l in lines[a - 1:b]:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."): continue
        c[cat(l.split()[0])] += 1
    valu = sum(c[k] for k in order[:9])
    print("%-28s %6d | " % (name, valu) + " ".join("%9d" % c[o] for o in order))
