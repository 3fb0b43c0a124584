#!/usr/bin/env python3
"""BUILD-CONTAINER-ONLY checker tool: a problem pair that carries the reference templates' LITERAL C.

The reference writes iLQG_problem.h / iLQG_func.c by running Maxima + gentran over iLQG_problem.tem / iLQG_func.tem
(make_iLQG.mac:7-8): every line of a template that is not inside <<...>> is copied to the output as it stands, every
<<...>> block is replaced by what its Maxima code prints.  Maxima is not in this image, so `tools/gen_problem.py` emits
pairs with function bodies of its own.  What that leaves open is whether the template's own text — limitsU() with its
index arrays, pointer walks and switch, forward_pass(), calc_derivs(), init_opt(), update_multipliers_*(), the
`#define aux_<name> t-><name>` lines — survives the device wrapper of ddp-generator_amd/csrc/ilqg_kernels.hip.

This tool answers it where the reference is present: it READS the two templates where they lie (default
/root/reference), keeps every literal line, and fills only the <<...>> blocks from gen_problem.py's expression printers
(the symbolic work is the same; the block -> printer map below follows genenerator_main.mac:189-447).  Its output is a
derivative of the reference's files: it goes to a directory that is neither tracked nor shipped (oracle/_ref/tem/, in
.gitignore and .gpurunignore) and is never committed.  Every block of a template must be recognised, else the tool
stops — a template that changed is not silently half-filled.

    python tools/fill_reference_template.py problems/defs/carparking.py oracle/_ref/tem/carparking_tem [--ref DIR]
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_problem as gp  # noqa: E402


class Filler:
    def __init__(self, em):
        self.em = em
        self.p = em.p
        self.unmatched = []

    # ---- helpers -----------------------------------------------------------
    def members(self, need, kinds):
        return "".join("    double %s;\n" % s.name for s in self.em.aux_syms(need, kinds))

    def run_members(self, kinds):
        full = self.members(self.em.run_need_full, kinds)
        return self.members(self.em.run_need, kinds) + ("#if FULL_DDP\n" + full + "#endif\n" if full else "")

    def defines(self, kinds):
        em = self.em
        out = ""
        for s in em.D.order:
            if em.D.kind[s] in kinds and (s in em.run_need or s in em.run_need_full or s in em.fin_need):
                out += "#define %s t->%s\n" % (em.macro_name(s), s.name)
        return out

    def mu_defines(self, kind):
        return "".join("#define mu_%s_%d m->mu_%s[%d]\n" % (kind, i + 1, kind, i) for i in range(len(self.em.cons_al[kind])))

    def mul_member(self, kind, what):
        n = len(self.em.cons_al[kind])
        return "    double %s_%s[%d];\n" % (what, kind, n) if what == "mu" and n else ("    double last_h%s[%d];\n" % (kind, n) if n else "")

    def running_aux(self, mode):
        """print_aux(mode, used_by_running): 0 constant, 1 time varying without u, 2 time varying with u"""
        em = self.em
        need = em.run_need | em.run_need_full
        if mode == 0:
            return em.aux_block(need, ("aux",), False, ind=4)
        return em.aux_block(need, ("aux",), True, u_dep=(mode == 2))

    def running_deriv(self, tv):
        em = self.em
        full = em.aux_block(em.run_need_full, ("d1", "d2"), tv)
        return em.aux_block(em.run_need, ("d1", "d2"), tv) + ("#if FULL_DDP\n" + full + "#endif\n" if full else "")

    def tensor(self, name, tv, zero_fill):
        em = self.em
        ten = getattr(em, name)
        if em.all_zero(ten):
            size = {"fxx": "sizeofQxx", "fuu": "sizeofQuu", "fxu": "sizeofQxu"}[name]
            return "    memset(t->%s, 0, sizeof(double)*N_X*%s);\n" % (name, size) if zero_fill else ""
        return em.block(em.jaco2_items(name, ten), tv)

    def clamp(self):
        out = ""
        for c in self.em.cons:
            j = c["input"]
            out += "// constraint h[%d]= %s\n" % (c["index"] + 1, gp.sp.sstr(c["expr"]))
            out += self.em.assign("limit", c["limit"], guard=False)
            out += "    if(u[%d]%slimit)\n        u[%d]= limit;\n\n" % (j, ">" if c["sign"] > 0 else "<", j)
        return out

    def limits(self):
        out = ""
        for c in self.em.cons:
            j = c["input"]
            arr, cmp_, idx = ("upper", ">", "upper_idx") if c["sign"] > 0 else ("lower", "<", "lower_idx")
            out += "// constraint h[%d]= %s\n" % (c["index"] + 1, gp.sp.sstr(c["expr"]))
            out += self.em.assign("limit", c["limit"], guard=False)
            out += "    if(t->%s[%d]%slimit) {\n        t->%s[%d]= limit;\n        %s[%d]= %d;\n    }\n\n" % (arr, j, cmp_, arr, j, idx, j, c["index"])
        return out

    def cases(self):
        out = ""
        for c in self.em.cons:
            out += "                case %d:\n// constraint h[%d]= %s\n" % (c["index"], c["index"] + 1, gp.sp.sstr(c["expr"]))
            for jx in range(self.em.n):
                out += self.em.assign("hx_[%d]" % jx, c["hx"][jx], ind=20, guard=False)
            out += "                    h_sign[0]= %d.0;\n                    break;\n" % c["sign"]
        return out

    def mul_init(self, kind):
        n = len(self.em.cons_al[kind])
        if not n:
            return ""
        pad = "        " if kind[0] == "l" else "    "
        return "%sfor(i= 0; i<%d; i++) { m->mu_%s[i]= %s; m->last_h%s[i]= 0.0; }\n" % (pad, n, kind, "0.0" if kind[1] == "e" else "1.0", kind)

    def progress(self, kind):
        """the stall test of every constraint of a kind and the violation it remembers (iLQG_func.tem:428-440, 471-483)"""
        ind = "        " if kind[0] == "l" else "    "
        out = ""
        for i, (hs, mu) in enumerate(self.em.cons_al[kind]):
            h = self.em.macro_name(hs)
            if kind[1] == "e":
                out += "%sif(fabs(%s)>o->tolConstraint && o->w_pen_fact1*fabs(%s)>fabs(m->last_h%s[%d]))\n%s    increase_pen= 1;\n" % (ind, h, h, kind, i, ind)
            else:
                out += "%sif(%s>o->tolConstraint && o->w_pen_fact1*%s>m->last_h%s[%d])\n%s    increase_pen= 1;\n" % (ind, h, h, kind, i, ind)
            out += "%sm->last_h%s[%d]= %s;\n\n" % (ind, kind, i, h)
        return out

    def mu_equality(self, kind):
        ind = 8 if kind[0] == "l" else 4
        w = self.em.w_pen
        return "".join(self.em.assign("mu_%s_%d" % (kind, i + 1), mu + w * hs, ind) for i, (hs, mu) in enumerate(self.em.cons_al[kind]))

    def mu_inequality(self, kind):
        ind = 4
        w = self.em.w_pen
        out = ""
        for i, (hs, mu) in enumerate(self.em.cons_al[kind]):
            lhs = "mu_%s_%d" % (kind, i + 1)
            out += "    if(%s>=0) {\n" % self.em.macro_name(hs)
            out += self.em.assign(lhs, mu * (1 + 2 * w * hs), ind + 4)
            out += "    } else {\n"
            out += self.em.assign(lhs, mu * (1 - w * hs) ** -2, ind + 4)
            out += "    }\n\n"
        return out

    # ---- block -> text ------------------------------------------------------
    def rules(self, which):
        em, p, n, m = self.em, self.p, self.em.n, self.em.m
        names = em.param_names
        stamp = "/* The reference's %s with its Maxima blocks filled by tools/fill_reference_template.py for '%s'.\n * Build-container checker output: not tracked, not shipped. */\n\n"
        if which == "problem":
            return [
                (r'^gentran\(literal\("\\/\\\* File generated', lambda: stamp % ("iLQG_problem.tem", p.name)),
                (r"^gentran\(eval\(nx\)\)\$$", lambda: str(n)),
                (r"^gentran\(eval\(nu\)\)\$$", lambda: str(m)),
                (r"^gentran\(eval\(\(nx\*\(nx\+1\)\)/2\)\)\$$", lambda: str(n * (n + 1) // 2)),
                (r"^gentran\(eval\(\(nu\*\(nu\+1\)\)/2\)\)\$$", lambda: str(m * (m + 1) // 2)),
                (r"^gentran\(eval\(nx\*nu\)\)\$$", lambda: str(n * m)),
                (r"for a in aux_def do if get\(a\[1\], used_by_running\)", lambda: self.run_members(("aux",))),
                (r"for a in aux_deriv do if get\(a\[1\], used_by_running\)", lambda: self.run_members(("d1", "d2"))),
                (r"for a in aux_def do if get\(a\[1\], used_by_final\)", lambda: self.members(em.fin_need, ("aux",))),
                (r"for a in aux_deriv do if get\(a\[1\], used_by_final\)", lambda: self.members(em.fin_need, ("d1", "d2"))),
            ] + [(r'^if n_h%s#0 then gentran\(literal\(" double %s' % (k, "mu_" if w == "mu" else "last_h"), (lambda k=k, w=w: self.mul_member(k, w)))
                 for k in ("le", "li", "fe", "fi") for w in ("mu", "last")]
        grad = {"Lx": ("cx", em.Lx), "Lu": ("cu", em.Lu), "Fx": ("cx", em.Fx)}
        hess = {"Lxx": ("cxx", em.Lxx), "Luu": ("cuu", em.Luu), "Lxu": ("cxu", em.Lxu), "Fxx": ("cxx", em.Fxx)}
        jaco = {"fx": em.fx, "fu": em.fu}
        R = [
            (r'^gentran\(literal\("\\/\\\* File generated', lambda: stamp % ("iLQG_func.tem", p.name)),
            (r"^tri_matrix_mode: true;$", lambda: ""),
            (r"^gentran\(eval\(length\(params\)\)\)\$$", lambda: str(len(names))),
            (r'for i:1 thru length\(params\) do gentran\(literal\("tParamDesc p_name"',
             lambda: "".join('tParamDesc p_name%d= {"%s", %d, 0};\n' % (i + 1, nm, p.params[nm]) for i, nm in enumerate(names))),
            (r'for i:1 thru length\(params\) do \(gentran\(literal\(&, "p_name"', lambda: ", ".join("&p_name%d" % (i + 1) for i in range(len(names)))),
            (r'^for a_work in aux_def do gentran\(literal\("#define "', lambda: self.defines(("aux",))),
            (r'^for a_work in aux_deriv do gentran\(literal\("#define "', lambda: self.defines(("d1", "d2"))),
        ]
        for k in ("fe", "fi", "le", "li"):
            R.append((r'thru n_h%s do gentran\(literal\("#define mu_%s_"' % (k, k), (lambda k=k: self.mu_defines(k))))
        R += [
            (r"^do_assign\('t\\-\\>c, L, 4, 0\)$", lambda: em.assign("t->c", p.L)),
            (r"^do_assign\('t\\-\\>c, F, 4, 0\)$", lambda: em.assign("t->c", p.F)),
            (r"for i:1 thru nx do do_assign\(x_next\[i-1\]", lambda: "".join(em.assign("x_next[%d]" % r, p.f[r]) for r in range(n))),
            (r"thru nh do do_clamp\(i\)", self.clamp),
            (r"thru nh do do_limits\('t\\-\\>, i\)", self.limits),
            (r"do_hx\('hx_, 'h_sign, i, 4\*5\)", self.cases),
            (r"^print_aux\(1, used_by_running\);$", lambda: self.running_aux(1)),
            (r"^print_aux\(2, used_by_running\);$", lambda: self.running_aux(2)),
            (r"^print_aux\(0, used_by_running\);$", lambda: self.running_aux(0)),
            (r"^print_aux\(1, used_by_final\);$", lambda: em.aux_block(em.fin_need, ("aux",), True)),
            (r"^print_aux\(0, used_by_final\);$", lambda: em.aux_block(em.fin_need, ("aux",), False)),
            (r"^print_deriv\(1, used_by_running\);$", lambda: self.running_deriv(True)),
            (r"^print_deriv\(0, used_by_running\);$", lambda: self.running_deriv(False)),
            (r"^print_deriv\(1, used_by_final\);$", lambda: em.aux_block(em.fin_need, ("d1", "d2"), True)),
            (r"^print_deriv\(0, used_by_final\);$", lambda: em.aux_block(em.fin_need, ("d1", "d2"), False)),
        ]
        for nm, mat in jaco.items():
            for flag, tv in (("true", True), ("false", False)):
                R.append((r"^print_jaco\('t\\-\\>%s, %s, %s\);$" % (nm, nm, flag), (lambda nm=nm, mat=mat, tv=tv: em.block(em.jaco_items(nm, mat), tv))))
        for nm in ("fxx", "fuu", "fxu"):
            R.append((r"^if not\(all_zero\(%s\)\) then print_jaco2\('t\\-\\>%s, %s, true\);$" % (nm, nm, nm), (lambda nm=nm: self.tensor(nm, True, False))))
            R.append((r"^if not\(all_zero\(%s\)\) then print_jaco2\('t\\-\\>%s, %s, false\) else gentran\(literal\(\" memset" % (nm, nm, nm),
                      (lambda nm=nm: self.tensor(nm, False, True))))
        for src, (dst, v) in grad.items():
            R.append((r"^print_grad\('t\\-\\>%s, %s\);$" % (dst, src), (lambda dst=dst, v=v: em.block(em.grad_items(dst, v), True))))
            R.append((r"^print_grad\('t\\-\\>%s, %s, false\);$" % (dst, src), (lambda dst=dst, v=v: em.block(em.grad_items(dst, v), False))))
        for src, (dst, v) in hess.items():
            R.append((r"^print_hess\('t\\-\\>%s, %s\);$" % (dst, src), (lambda dst=dst, v=v: em.block(em.hess_items(dst, v), True))))
            R.append((r"^print_hess\('t\\-\\>%s, %s, false\);$" % (dst, src), (lambda dst=dst, v=v: em.block(em.hess_items(dst, v), False))))
        for k in ("le", "li", "fe", "fi"):
            R.append((r'^if n_h%s#0 then gentran\(literal\(" +for\(i= 0\\; i<"' % k, (lambda k=k: self.mul_init(k))))
            R.append((r"thru n_h%s do \( gentran\(if " % k, (lambda k=k: self.progress(k))))
        R += [
            (r"thru n_hle do do_assign\(concat\(mu_le_, i\), mu_le_next\[i\]", lambda: self.mu_equality("le")),
            (r"thru n_hfe do do_assign\(concat\(mu_fe_, i\), mu_fe_next\[i\]", lambda: self.mu_equality("fe")),
            (r"thru n_hli do \( gentran\(literal\(\" if\(\"", lambda: self.mu_inequality("li")),
            (r"thru n_hfi do \( gentran\(literal\(\" if\(\"", lambda: self.mu_inequality("fi")),
            (r'^if member\(g, arrays\) then gentran\(literal\("return\("', lambda: "    return(0);\n"),
            (r"^block\(\[i, i_: 0\], if member\(g, arrays\)", lambda: ""),
        ]
        return R

    def fill(self, text, which):
        rules = [(re.compile(pat), fn) for pat, fn in self.rules(which)]
        used = set()

        def sub(mo):
            body = re.sub(r"\s+", " ", mo.group(1)).strip()
            hits = [i for i, (pat, _) in enumerate(rules) if pat.search(body)]
            if len(hits) != 1:
                self.unmatched.append((which, body[:100], len(hits)))
                return mo.group(0)
            used.add(hits[0])
            return rules[hits[0]][1]()

        out = re.sub(r"<<(.*?)>>", sub, text, flags=re.S)
        # a block that stands alone on its line(s) leaves nothing but its output
        out = re.sub(r"\n[ \t]+\n", "\n\n", out)
        return out


def literal_lines(template_text):
    """the template's own lines: what is left of it once the <<...>> blocks are taken out (blank lines dropped)"""
    bare = re.sub(r"<<.*?>>", "\x00", template_text, flags=re.S)
    out = []
    for line in bare.split("\n"):
        for piece in line.split("\x00"):  # a block in the middle of a line: the text on either side of it
            if piece.strip():
                out.append(piece.strip())
    return out


def missing_literals(template_text, filled_text):
    """literal pieces of the template that are NOT in the filled file, searched in order (each behind the one before)"""
    pos, missing = 0, []
    for piece in literal_lines(template_text):
        at = filled_text.find(piece, pos)
        if at < 0:
            missing.append(piece)
        else:
            pos = at + len(piece)
    return missing


def main(argv):
    ref = "/root/reference"
    if "--ref" in argv:
        i = argv.index("--ref")
        ref = argv[i + 1]
        del argv[i:i + 2]
    if len(argv) != 3:
        print(__doc__)
        return 2
    tems = {w: os.path.join(ref, "iLQG_%s.tem" % w) for w in ("problem", "func")}
    if not all(os.path.exists(t) for t in tems.values()):
        print("fill_reference_template: no templates under %s (build-container-only tool): nothing written" % ref)
        return 3
    prob = gp.load_problem(argv[1])
    em = gp.Emitter(prob, plain=True)  # no additive hints or tables: what Maxima would have to work with
    f = Filler(em)
    outs = {w: f.fill(open(t).read(), w) for w, t in tems.items()}
    if f.unmatched:
        for which, body, hits in f.unmatched:
            print("fill_reference_template: iLQG_%s.tem block matched by %d rules: %s" % (which, hits, body), file=sys.stderr)
        return 1
    for w, t in tems.items():
        lost = missing_literals(open(t).read(), outs[w])
        if lost:
            print("fill_reference_template: literal text of iLQG_%s.tem lost: %r" % (w, lost[:5]), file=sys.stderr)
            return 1
    os.makedirs(argv[2], exist_ok=True)
    for w, name in (("problem", "iLQG_problem.h"), ("func", "iLQG_func.c")):
        path = os.path.join(argv[2], name)
        if os.path.exists(path) and open(path).read() == outs[w]:
            continue  # (unchanged: keep the file's time, so that make rebuilds nothing that depends on it)
        with open(path, "w") as fh:
            fh.write(outs[w])
    print("wrote %s/{iLQG_problem.h,iLQG_func.c} from the templates under %s" % (argv[2], ref))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
