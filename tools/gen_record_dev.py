#!/usr/bin/env python3
"""Build step of the device wrapper (ddp-generator_amd/csrc/Makefile): reads a generated problem pair
{iLQG_problem.h, iLQG_func.c} — the reference's templates iLQG_problem.tem:23-51 / iLQG_func.tem:262-347 fix their form —
and writes ilqg_record_dev.h, which tells ilqg_kernels.hip

  * the members of trajEl_t in order (ILQG_DEV_MEMBERS): the kernels of the wave mapping hand the generated callbacks a
    PRIVATE element per lane whose derivative arrays (cx .. fxu, 47 KB of the n = 16 problem's 47.9 KB) are not storage
    but proxies — an assignment `t->fxx[17]= ...` of the unmodified function file puts the value into a ring in LDS;
  * the ORDER in which init_running and bp_derivsL assign those entries (ILQG_DEV_SEQ_INIT / ILQG_DEV_SEQ_DERIVS): runs of
    neighbouring entries are written out by the whole wavefront, 512 contiguous bytes of one record per store
    instruction, when their last entry has been assigned — which entry that is, is a compile-time table made from
    this order.

Nothing of the pair is changed or copied.  The pair is only used this way (ILQG_DEV_RECORDS 1) if the scan can vouch
for it: every access to a derivative member anywhere in the function file is an assignment with a literal index inside
init_running / bp_derivsL, the guard behind it (isNANorINF / PRNT of the entry just assigned) or a memset of a whole member
to zero inside init_running; no preprocessor conditional other than `#if FULL_DDP` inside those functions or the struct;
runs long enough to pay.  Otherwise ILQG_DEV_RECORDS is 0, with the reason, and the kernels store entry by entry as before.

    gen_record_dev.py <problem dir> <FULL_DDP> <out.h>
"""
import re
import sys

PROXIED = ("cx", "cxx", "cu", "cuu", "cxu", "fx", "fu", "fxx", "fuu", "fxu")
MIN_MEAN_RUN = 8.0


class Unsupported(Exception):
    pass


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def resolve_conditionals(lines, fd, where):
    """lines with `#if FULL_DDP` / `#if !FULL_DDP` / `#else` / `#endif` resolved; any other directive: Unsupported"""
    out, stack = [], []
    for ln in lines:
        s = ln.strip()
        if s.startswith("#"):
            d = re.sub(r"\s+", " ", s[1:].strip())
            if d in ("if FULL_DDP", "if FULL_DDP!=0", "if FULL_DDP != 0", "ifdef FULL_DDP") or d == "if (FULL_DDP)":
                stack.append(bool(fd))
            elif d in ("if !FULL_DDP", "if FULL_DDP==0", "if FULL_DDP == 0"):
                stack.append(not fd)
            elif d == "else" and stack:
                stack[-1] = not stack[-1]
            elif d.startswith("endif") and stack:
                stack.pop()
            else:
                raise Unsupported("preprocessor directive `%s` inside %s" % (s, where))
            continue
        if all(stack):
            out.append(ln)
    if stack:
        raise Unsupported("unbalanced #if inside %s" % where)
    return out


def struct_members(header, fd):
    """[(name, count expression or None for a scalar)] of trajEl_t, in order"""
    h = strip_comments(header)
    m = re.search(r"typedef\s+struct\s*(?:\w+\s*)?\{(.*?)\}\s*trajEl_t\s*;", h, flags=re.S)
    if not m:
        raise Unsupported("no `typedef struct { ... } trajEl_t;` in iLQG_problem.h")
    members = []
    for ln in resolve_conditionals(m.group(1).splitlines(), fd, "trajEl_t"):
        for decl in filter(None, (d.strip() for d in ln.split(";"))):
            mm = re.fullmatch(r"double\s+(\w+)\s*(?:\[(.+)\])?", decl)
            if not mm:
                raise Unsupported("member of trajEl_t that is not `double name;` or `double name[count];`: `%s`" % decl)
            members.append((mm.group(1), mm.group(2)))
    return members


def defines(header):
    """the header's integer #defines (N_X, N_U, sizeofQxx, ...), evaluated"""
    env = {}
    for name, val in re.findall(r"^\s*#\s*define\s+(\w+)\s+([^\n/]+?)\s*(?://.*)?$", strip_comments(header), flags=re.M):
        try:
            env[name] = int(eval(val, {"__builtins__": {}}, dict(env)))
        except Exception:
            pass
    return env


def function_body(src, name):
    """text between the braces of `static int <name>(trajEl_t *t, ...) {`"""
    m = re.search(r"\bstatic\s+int\s+%s\s*\(\s*trajEl_t\s*\*\s*t\b[^)]*\)\s*\{" % name, src)
    if not m:
        raise Unsupported("no definition `static int %s(trajEl_t *t, ...)` in iLQG_func.c" % name)
    depth, i = 1, m.end()
    while depth and i < len(src):
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    if depth:
        raise Unsupported("unbalanced braces in %s" % name)
    return m.end(), i - 1


def scan(problem_dir, fd):
    header = open(problem_dir + "/iLQG_problem.h").read()
    src = strip_comments(open(problem_dir + "/iLQG_func.c").read())
    members = struct_members(header, fd)
    names = [n for n, _ in members]
    proxied = [n for n in PROXIED if n in names]
    if not proxied:
        raise Unsupported("trajEl_t has none of the derivative members")
    env = defines(header)
    env["FULL_DDP"] = fd
    count = {}
    for n, c in members:
        if c is not None:
            try:
                count[n] = int(eval(c, {"__builtins__": {}}, dict(env)))
            except Exception:
                raise Unsupported("size of member %s: `%s`" % (n, c))
    member_re = "|".join(proxied)
    assign = re.compile(r"^[ \t]*t->(%s)\[\s*(\d+)\s*\]\s*=(?!=)" % member_re, flags=re.M)
    seqs, consumed = {}, []
    for fn in ("init_running", "bp_derivsL"):
        a, b = function_body(src, fn)
        body = "\n".join(resolve_conditionals(src[a:b].splitlines(), fd, fn))
        seq = []
        pos = 0
        # assignments and (init_running) memsets of whole members, in textual order
        token = re.compile(r"^[ \t]*t->(%s)\[\s*(\d+)\s*\]\s*=(?!=)|memset\s*\(\s*t->(%s)\s*,\s*0\s*,\s*sizeof\s*\(\s*double\s*\)\s*\*([^;]*?)\)\s*;"
                           % (member_re, member_re), flags=re.M)
        for m in token.finditer(body):
            if m.group(1):
                seq.append((m.group(1), int(m.group(2))))
            else:
                if fn != "init_running":
                    raise Unsupported("memset of a derivative member outside init_running")
                n = m.group(3)
                try:
                    cnt = int(eval(m.group(4), {"__builtins__": {}}, dict(env)))
                except Exception:
                    raise Unsupported("memset size `%s`" % m.group(4))
                if cnt != count[n]:
                    raise Unsupported("memset of part of member %s" % n)
                seq.extend((n, i) for i in range(cnt))
        for n, i in seq:
            if i >= count[n]:
                raise Unsupported("%s[%d] is beyond the member" % (n, i))
        seqs[fn] = seq
        # what is left of the body once assignments' left-hand sides, their guards and the memsets are taken out
        rest = re.sub(r"^[ \t]*if\s*\(\s*isNANorINF\s*\(\s*t->(?:%s)\[\s*\d+\s*\]\s*\)\s*\)\s*\{[^}\n]*\}[ \t]*$" % member_re, "", body, flags=re.M)
        rest = token.sub(lambda m: "" if m.group(3) else "=", rest)
        consumed.append((a, b, rest))
    # every other mention of a derivative member of a trajEl_t, anywhere
    outside = src
    for a, b, rest in sorted(consumed, reverse=True):
        outside = outside[:a] + rest + outside[b:]
    # (trajFin_t has cx and cxx of its own: `t->cx` inside functions of the final element is not ours)
    for fn in ("bp_derivsF", "init_final", "calcFVariableAux", "calcFAuxDeriv", "ddpF"):
        m = re.search(r"\b(?:static\s+)?int\s+%s\s*\(\s*trajFin_t\s*\*\s*t\b[^)]*\)\s*\{" % fn, outside)
        if m:
            depth, i = 1, m.end()
            while depth and i < len(outside):
                depth += {"{": 1, "}": -1}.get(outside[i], 0)
                i += 1
            outside = outside[:m.end()] + outside[i - 1:]
    left = re.search(r"(?:->|\.)\s*(%s)\b(?!\s*\()" % member_re, outside)
    if left:
        line = outside.count("\n", 0, left.start()) + 1
        # (members of the final element reached through `f.` / `->f.`: cx, cxx of trajFin_t)
        ctx = outside[max(0, left.start() - 12):left.start()]
        if not re.search(r"(?:\bf|->f|\.f)\s*$", ctx):
            raise Unsupported("derivative member `%s` is accessed outside an assignment of init_running / bp_derivsL (iLQG_func.c line %d)"
                              % (left.group(1), line))
    both = set(seqs["init_running"]) & set(seqs["bp_derivsL"])
    if both:
        raise Unsupported("%d entries are assigned by init_running AND bp_derivsL (e.g. %s[%d])" % ((len(both),) + sorted(both)[0]))
    for fn, seq in seqs.items():
        if len(set(seq)) != len(seq):
            raise Unsupported("%s assigns an entry twice" % fn)
    # runs (neighbours in the record, cut at multiples of 64 entries): long enough?
    off, o = {}, 0
    for n, c in members:
        off[n] = o
        o += 1 if c is None else count[n]
    seq = seqs["bp_derivsL"]
    if not seq:
        raise Unsupported("bp_derivsL assigns no derivative entry")
    es = [off[n] + i for n, i in seq]
    runs = 1 + sum(1 for p, q in zip(es, es[1:]) if q != p + 1 or q % 64 == 0)
    if len(es) / runs < MIN_MEAN_RUN:
        raise Unsupported("bp_derivsL assigns its %d entries in %d runs of neighbours: too short to pay" % (len(es), runs))
    return members, proxied, seqs, (len(es), runs)


def limits_state_free(problem_dir):
    """True if limitsU() provably stores nothing but zeros as the limits' gradients (lower_hx / upper_hx): every assignment
    in its body to an indexed name that is not t->lower / t->upper, a sign or an index array has the literal 0 on its
    right — and there is at least one (the reference's do_hx prints one per state, genenerator_main.mac:419-447).  The
    backward kernels then leave signs and gradients alone (back_pass.c:186-199 multiplies by them: zero rows of K either
    way) and the records written for them need not carry either."""
    src = strip_comments(open(problem_dir + "/iLQG_func.c").read())
    m = re.search(r"\bstatic\s+void\s+limitsU\s*\(\s*trajEl_t\s*\*\s*t\b[^)]*\)\s*\{", src)
    if not m:
        return False
    depth, i = 1, m.end()
    while depth and i < len(src):
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    body = src[m.end():i - 1]
    if re.search(r"^\s*#", body, flags=re.M):
        return False
    grads = 0
    for lhs, idx, op, rhs in re.findall(r"([A-Za-z_][\w>.-]*)\s*\[([^\]]*)\]\s*([-+*/]?=)(?!=)\s*([^;]*);", body):
        name = lhs.split("->")[-1]
        if name in ("lower", "upper") or "sign" in name or "idx" in name or name == "active":
            continue
        if "hx" in name or "grad" in name:
            grads += 1
            if op != "=" or not re.fullmatch(r"[-+]?0(\.0*)?", rhs.strip()):
                return False
        else:
            return False  # an array this rule does not know
    # (gradients reached without an index: `*hx_++ = ...`, memcpy: not the printers' form)
    if re.search(r"\*\s*\w*hx\w*\s*(\+\+)?\s*=", body):
        return False
    return grads > 0


def main():
    problem_dir, fd, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    lines = ["/* Written by tools/gen_record_dev.py from the generated pair in %s (FULL_DDP=%d); see there. */" % (problem_dir, fd)]
    try:
        free = limits_state_free(problem_dir)
    except Exception:
        free = False
    lines.append("/* limitsU() stores zeros as the limits' gradients, and nothing else: %s */" % ("yes" if free else "not shown"))
    lines.append("#define ILQG_DEV_LIMITS_STATE_FREE %d" % (1 if free else 0))
    try:
        members, proxied, seqs, (n, runs) = scan(problem_dir, fd)
        lines.append("#define ILQG_DEV_RECORDS 1")
        lines.append("/* bp_derivsL: %d entries in %d runs; init_running: %d entries */" % (n, runs, len(seqs["init_running"])))
        parts = []
        for name, cnt in members:
            if name in proxied:
                parts.append("PROXY(%s, (%s))" % (name, cnt))
            elif cnt is None:
                parts.append("SCALAR(%s)" % name)
            else:
                parts.append("ARRAY(%s, (%s))" % (name, cnt))
        lines.append("#define ILQG_DEV_MEMBERS(SCALAR, ARRAY, PROXY) " + " ".join(parts))
        for macro, fn in (("ILQG_DEV_SEQ_INIT", "init_running"), ("ILQG_DEV_SEQ_DERIVS", "bp_derivsL")):
            lines.append("#define %s(E) %s" % (macro, " ".join("E(%s,%d)" % e for e in seqs[fn])))
    except Unsupported as e:
        lines.append("#define ILQG_DEV_RECORDS 0")
        lines.append("/* %s */" % str(e).replace("*/", "* /"))
    text = "\n".join(lines) + "\n"
    try:
        if open(out).read() == text:
            return
    except OSError:
        pass
    with open(out, "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
