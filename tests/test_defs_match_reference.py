"""Build container only: the problem definitions the goldens are made from (problems/defs/*.py — the reference commits no
generated files, and Maxima is absent: SURVEY §8c) against the reference's OWN definitions of the same problems,
examples/*/optDef*.mac.  The .mac files are read as text by the small Maxima-subset parser below (assignments `name: expr;`,
`f[x]: expr;`, function definitions `g(a, b):= expr;`, infix + - * / ^, calls, `'s` nouns, `[..]` lists and indices,
assume(), integrate()/factor()/expand()) and every expression — f, L, F, h, hfe, hli, the auxiliaries substituted — must
equal ours: sympy.simplify(a - b) == 0.  Nothing of the .mac text is stored; the test skips where /root/reference is absent
(the GPU box)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")


# ---- a Maxima subset ------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+|\d+)|([A-Za-z_%][A-Za-z_0-9%]*)|(:=|[-+*/^()\[\],:;$']))")


def tokens(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"assume\([^)]*\)\s*[;$]", " ", text)  # (read by read_assumptions)
    out, i = [], 0
    while i < len(text):
        m = TOKEN.match(text, i)
        if not m:
            if text[i:].strip() == "":
                break
            raise SyntaxError("cannot read %r" % text[i:i + 20])
        i = m.end()
        out.append(("num", m.group(1)) if m.group(1) else ("id", m.group(2)) if m.group(2) else ("op", m.group(3)))
    return out


class Parser:
    """expressions as nested tuples: ("num", s) ("sym", name) ("idx", name, expr) ("call", name, [args]) ("list", [..])
    ("neg", a) ("+", a, b) ..."""

    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else ("end", "")

    def take(self, kind=None, val=None):
        k, v = self.peek()
        if (kind and k != kind) or (val and v != val):
            raise SyntaxError("expected %s %s, got %s %s" % (kind, val, k, v))
        self.i += 1
        return v

    def statements(self):
        out = []
        while self.peek()[0] != "end":
            lhs = self.expr()
            k, v = self.peek()
            if (k, v) in (("op", ":"), ("op", ":=")):
                self.take()
                out.append((v, lhs, self.expr()))
            else:
                out.append(("do", lhs, None))
            if self.peek() in (("op", ";"), ("op", "$")):
                self.take()
        return out

    def expr(self):
        a = self.term()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.take()
            a = (op, a, self.term())
        return a

    def term(self):
        a = self.unary()
        while self.peek() in (("op", "*"), ("op", "/")):
            op = self.take()
            a = (op, a, self.unary())
        return a

    def unary(self):
        if self.peek() == ("op", "-"):
            self.take()
            return ("neg", self.unary())
        if self.peek() == ("op", "+"):
            self.take()
            return self.unary()
        return self.power()

    def power(self):
        a = self.atom()
        if self.peek() == ("op", "^"):
            self.take()
            return ("^", a, self.unary())  # right associative, binds tighter than unary minus on its left
        return a

    def atom(self):
        k, v = self.peek()
        if k == "num":
            self.take()
            return ("num", v)
        if (k, v) == ("op", "'"):  # a noun: the symbol itself
            self.take()
            return self.atom()
        if (k, v) == ("op", "("):
            self.take()
            a = self.expr()
            self.take("op", ")")
            return a
        if (k, v) == ("op", "["):
            self.take()
            items = [self.expr()]
            while self.peek() == ("op", ","):
                self.take()
                items.append(self.expr())
            self.take("op", "]")
            return ("list", items)
        if k == "id":
            self.take()
            if self.peek() == ("op", "("):
                self.take()
                args = []
                if self.peek() != ("op", ")"):
                    args.append(self.expr())
                    while self.peek() == ("op", ","):
                        self.take()
                        args.append(self.expr())
                self.take("op", ")")
                return ("call", v, args)
            if self.peek() == ("op", "["):
                self.take()
                ix = self.expr()
                self.take("op", "]")
                return ("idx", v, ix)
            return ("sym", v)
        raise SyntaxError("unexpected %s %s" % (k, v))


def read_mac(path):
    """{"x": [names], "u": [names], "aux": {name: tree}, "f": {state: tree}, "L", "F", "h": {i: tree}, "hfe", "hli", ...,
    "functions": {name: (params, tree)}, "negative": set, "positive": set}"""
    out = {"aux": {}, "f": {}, "h": {}, "hfe": {}, "hli": {}, "hle": {}, "hfi": {}, "functions": {}, "negative": set(), "positive": set()}
    for op, lhs, rhs in Parser(tokens(open(path).read())).statements():
        if op == "do":
            continue
        if op == ":=":
            assert lhs[0] == "call"
            out["functions"][lhs[1]] = ([a[1] for a in lhs[2]], rhs)
        elif lhs[0] == "sym" and lhs[1] in ("x", "u"):
            out[lhs[1]] = [e[1] for e in rhs[1]]
        elif lhs[0] == "sym" and lhs[1] in ("L", "F"):
            out[lhs[1]] = rhs
        elif lhs[0] == "sym":
            out["aux"][lhs[1]] = rhs
        elif lhs[0] == "idx" and lhs[1] == "f":
            out["f"][lhs[2][1]] = rhs
        elif lhs[0] == "idx" and lhs[1] in ("h", "hfe", "hli", "hle", "hfi"):
            out[lhs[1]][int(lhs[2][1])] = rhs
        else:
            raise SyntaxError("statement %r" % (lhs,))
    return out


def read_assumptions(path):
    """assume(name<0) / assume(name>0) of the file: {name: "negative" | "positive"}"""
    text = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
    return {m.group(1): ("negative" if m.group(2) == "<" else "positive") for m in re.finditer(r"assume\(\s*(\w+)\s*([<>])\s*0\s*\)", text)}


def to_sympy(tree, sym, functions):
    import sympy as sp
    k = tree[0]
    if k == "num":
        return sp.Rational(tree[1]) if re.fullmatch(r"\d+", tree[1]) else sp.Float(tree[1])
    if k == "sym":
        return sym(tree[1])
    if k == "idx":
        ix = tree[2]
        return sym("%s[%s]" % (tree[1], ix[1]))
    if k == "neg":
        return -to_sympy(tree[1], sym, functions)
    if k in "+-*/^":
        a, b = to_sympy(tree[1], sym, functions), to_sympy(tree[2], sym, functions)
        return {"+": a + b, "-": a - b, "*": a * b, "/": a / b, "^": a ** b}[k]
    if k == "call":
        name, args = tree[1], tree[2]
        if name in functions:
            params, body = functions[name]
            bound = dict(zip(params, [to_sympy(a, sym, functions) for a in args]))
            return to_sympy(body, lambda n: bound[n] if n in bound else sym(n), functions)
        if name in ("factor", "expand"):
            return to_sympy(args[0], sym, functions)
        if name == "integrate":
            f, v, a, b = (to_sympy(t, sym, functions) for t in args)
            return sp.integrate(f, (v, a, b))
        fn = {"sin": sp.sin, "cos": sp.cos, "asin": sp.asin, "sqrt": sp.sqrt, "abs": sp.Abs, "exp": sp.exp, "log": sp.log, "tan": sp.tan,
              "atan": sp.atan, "acos": sp.acos}[name]
        return fn(*[to_sympy(a, sym, functions) for a in args])
    raise SyntaxError(str(tree))


def ours(name):
    """problems/defs/<name>.py through the generator's own Problem class"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_problem
    return gen_problem.load_problem(os.path.join(ROOT, "problems", "defs", name + ".py"))


@pytest.mark.parametrize("mac,defs", [("CarParking/optDefCar.mac", "carparking"), ("Brachistochrone/optDefBrachi.mac", "brachi"),
                                      ("Brachistochrone/optDefBrachi_hli.mac", "brachi_hli")])
def test_definition_equals_the_reference_mac(mac, defs):
    import sympy as sp
    path = os.path.join(REF, mac)
    M = read_mac(path)
    P = ours(defs)
    assumed = read_assumptions(path)
    # the integration variable of the Brachistochrone's cost runs over [0, dx]: positive
    table = {}

    def sym(name):
        if name not in table:
            kind = assumed.get(name, "positive" if name == "x_" and "integrate" in open(path).read() else None)
            table[name] = sp.Symbol(name, real=True, **({kind: True} if kind else {}))
        return table[name]

    assert [str(s) for s in P.x] == M["x"] and [str(s) for s in P.u] == M["u"]
    aux_theirs = {sym(n): to_sympy(t, sym, M["functions"]) for n, t in M["aux"].items()}
    rename = lambda e: sp.sympify(e).xreplace({s: sym(str(s)) for s in sp.sympify(e).free_symbols})
    aux_ours = {sym(str(s)): rename(d) for s, d in P.aux}

    def same(theirs, mine, what):
        a = to_sympy(theirs, sym, M["functions"]).subs(aux_theirs)
        b = rename(mine).subs(aux_ours)
        d = sp.simplify(a - b)
        assert d == 0, "%s of %s differs from %s: %s" % (what, defs, mac, d)

    assert set(M["aux"]) == {str(s) for s, _ in P.aux}
    for (s, d) in P.aux:
        same(M["aux"][str(s)], d, "auxiliary " + str(s))
    assert set(M["f"]) == set(M["x"])
    for i, xn in enumerate(M["x"]):
        same(M["f"][xn], P.f[i], "f[%s]" % xn)
    same(M["L"], P.L, "L")
    same(M["F"], P.F, "F")
    for kind in ("h", "hfe", "hli", "hle", "hfi"):
        mine = getattr(P, kind)
        assert sorted(M[kind]) == list(range(1, len(mine) + 1)), kind
        for i, e in enumerate(mine):
            same(M[kind][i + 1], e, "%s[%d]" % (kind, i + 1))
