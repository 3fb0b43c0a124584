#!/usr/bin/env python3
"""Emit iLQG_problem.h / iLQG_func.c for a symbolic optimal-control problem.

The reference generates these two files with Maxima + gentran from the
templates iLQG_problem.tem / iLQG_func.tem (driver make_iLQG.mac:7-8,
printers genenerator_main.mac:204-447).  Maxima is not available in this
environment and the reference does not commit generated files
(.gitignore:36-37), so this tool produces conforming files with sympy.  It is
a build-time tool only; nothing on the solver path imports it.

What "conforming" means (layout contract, all from the reference):
  * iLQG_problem.h: N_X, N_U, sizeofQxx/Quu/Qxu, trajEl_t / trajFin_t member
    order, traj_t, multiplier structs           (iLQG_problem.tem:16-89)
  * array layouts: fx[r+c*n], packed upper-tri hessians, fxx[r*sxx+UTRI(c,d)],
    fxu[r*n*m + d*n + c]                        (genenerator_main.mac:204-281)
  * function names, signatures and the order of operations inside
    forward_pass / calc_derivs / limitsU / clampU / init_opt
                                                (iLQG_func.tem:40-521)
  * every non-constant assignment is followed by a NaN/Inf guard that prints
    and returns 0                               (genenerator_main.mac:189-199)
  * constant derivative entries are written once by init_opt, time-varying
    ones by calc_derivs                         (iLQG_func.tem:262-347)
  * auxiliary variables and their derivatives are struct members evaluated
    once per step and reused                    (gen_dep_graph.mac:187-229)

Augmented-Lagrangian constraints hle / hli (running) and hfe / hfi (final) become penalty terms of
L resp. F with multipliers mu and the penalty weight w_pen, exactly as genenerator_main.mac:46-124
builds them (equality: mu*h + 0.5*w_pen*h^2; inequality after Ruxton: mu*h*(1+w_pen*h) for h >= 0,
mu*h/(1-w_pen*h) else); multiplier structs, init_multipliers and update_multipliers follow
iLQG_problem.tem:70-89 and iLQG_func.tem:371-509.

Usage:  python tools/gen_problem.py problems/defs/carparking.py problems/carparking
"""
import importlib.util
import itertools
import os
import re
import sys

import sympy as sp
from sympy.printing.c import C99CodePrinter


# --------------------------------------------------------------------------
# problem description
# --------------------------------------------------------------------------
class Problem:
    """Symbolic problem: states, inputs, parameters, auxiliaries, f, L, F, h."""

    def __init__(self, name):
        self.name = name
        self.x = []
        self.u = []
        self.params = {}  # name -> size (1 scalar, >1 vector, -1 per time step)
        self.aux = []     # list of (symbol, definition)
        self.f = None
        self.L = None
        self.F = None
        self.h = []       # input constraints h_i < 0
        # augmented-Lagrangian constraints (genenerator_main.mac:46-124): running equality / inequality (<= 0),
        # final equality / inequality; may depend on x (running ones also on u) and parameters
        self.hle, self.hli, self.hfe, self.hfi = [], [], [], []
        self.fast = False # True: skip sympy.simplify (large generated problems)
        self.cse = False  # True: name the products shared by derivative entries (SharedTerms), factored tensor tables

    def states(self, names):
        self.x = list(sp.symbols(names, real=True, seq=True))
        return self.x

    def inputs(self, names):
        self.u = list(sp.symbols(names, real=True, seq=True))
        return self.u

    def scalar(self, name):
        self.params[name] = 1
        return sp.Symbol(name, real=True)

    def vector(self, name, size):
        self.params[name] = size
        return [sp.Symbol("%s[%d]" % (name, i), real=True) for i in range(size)]

    def per_step(self, name):
        """parameter with one value per time step, referenced as name[k]"""
        self.params[name] = -1
        return sp.Symbol("%s[k]" % name, real=True)

    def auxiliary(self, name, definition):
        s = sp.Symbol(name, real=True)
        self.aux.append((s, sp.sympify(definition)))
        return s


# --------------------------------------------------------------------------
# C printing
# --------------------------------------------------------------------------
class _Printer(C99CodePrinter):
    """Prints small rational powers as products / sqrt so host libm and the
    device math library see the same cheap operations."""

    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Rational:
            num, den = abs(e.p), e.q
            if den in (1, 2) and num <= 8:
                bs = self._print(b)
                if not (b.is_Symbol or b.is_Function):
                    bs = "(" + bs + ")"
                factors = []
                if den == 2:
                    whole, half = divmod(num, 2)
                    factors += [bs] * whole
                    if half:
                        factors.append("sqrt(%s)" % self._print(b))
                else:
                    factors += [bs] * num
                body = "*".join(factors)
                if len(factors) > 1:
                    body = "(" + body + ")"
                return body if e.p > 0 else "1.0/" + body
        return super()._print_Pow(expr)

    def _print_Integer(self, expr):
        return "%d.0" % int(expr)

    def _print_Rational(self, expr):
        return "%d.0/%d.0" % (expr.p, expr.q)


_printer = _Printer({"precision": 17})


def cexpr(e):
    return _printer.doprint(e)


def _tidy(e):
    """simplify, then undo the multiple-angle forms simplify likes to introduce (sin(2w), cos(2w)):
    every distinct trig argument costs a separate argument reduction at run time"""
    e = sp.simplify(e)
    if any(isinstance(a, (sp.sin, sp.cos)) and a.args[0].is_Mul and a.args[0].as_coeff_Mul()[0] != 1
           for a in e.atoms(sp.sin, sp.cos)):
        e = sp.expand_trig(e)
    return e


def utri(r, c):
    return c * (c + 1) // 2 + r


# --------------------------------------------------------------------------
# derivative engine with auxiliary variables (chain rule through symbols)
# --------------------------------------------------------------------------
class Deriver:
    """Total derivatives of expressions that contain auxiliary symbols.

    Each auxiliary A has a definition def(A) in terms of states, inputs,
    parameters and earlier auxiliaries.  d A / d z is represented by a new
    symbol A_z whose definition is the total derivative of def(A); second
    derivatives likewise.  This mirrors the reference's gradef-based
    auxiliary derivatives (gen_dep_graph.mac:187-229).
    """

    def __init__(self, prob):
        self.prob = prob
        self.base = list(prob.x) + list(prob.u)
        self.base_name = {}
        for i, s in enumerate(prob.x):
            self.base_name[s] = "x%d" % i
        for i, s in enumerate(prob.u):
            self.base_name[s] = "u%d" % i
        self.defs = {}       # dependent symbol -> definition
        self.order = []      # evaluation order of dependent symbols
        self.kind = {}       # symbol -> 'aux' | 'd1' | 'd2'
        self.dcache = {}     # (symbol, z) -> derivative symbol or 0
        self._tags = {}      # derivative symbol -> (aux name, sorted variable tags)
        for s, d in prob.aux:
            self.defs[s] = d
            self.order.append(s)
            self.kind[s] = "aux"

    def total_diff(self, e, z):
        r = sp.diff(e, z)
        for s in sorted((s for s in e.free_symbols if s in self.defs), key=lambda q: q.name):  # sets have no order
            pd = sp.diff(e, s)
            if pd != 0:
                ds = self.dsym(s, z)
                if ds != 0:
                    r += pd * ds
        return r

    def dsym(self, s, z):
        key = (s, z)
        if key in self.dcache:
            return self.dcache[key]
        # canonical name: aux name + sorted variable tags
        d = self.total_diff(self.defs[s], z)
        if not self.prob.fast:
            d = _tidy(d)
        if d == 0:
            self.dcache[key] = sp.Integer(0)
            return self.dcache[key]
        if d.is_number:  # constant derivative: use the number itself, no struct member
            self.dcache[key] = d
            return d
        root, tags = self._root_tags(s)
        tags = sorted(tags + [self.base_name[z]])
        name = "d%s_%s" % (root, "".join(tags))
        existing = [q for q in self.defs if q.name == name]
        if existing:  # symmetric second derivative already created
            self.dcache[key] = existing[0]
            return existing[0]
        ns = sp.Symbol(name, real=True)
        self.defs[ns] = d
        self.order.append(ns)
        self.kind[ns] = "d1" if self.kind[s] == "aux" else "d2"
        self._tags[ns] = (root, tags)
        self.dcache[key] = ns
        return ns

    def _root_tags(self, s):
        if s in self._tags:
            r, t = self._tags[s]
            return r, list(t)
        return s.name, []

    def depends_on(self, e, syms):
        """does e depend (transitively through auxiliaries) on any of syms"""
        todo = list(e.free_symbols)
        seen = set()
        while todo:
            q = todo.pop()
            if q in seen:
                continue
            seen.add(q)
            if q in syms:
                return True
            if q in self.defs:
                todo += list(self.defs[q].free_symbols)
        return False

    def closure(self, exprs):
        """all dependent symbols needed to evaluate exprs, in evaluation order"""
        need = set()
        todo = []
        for e in exprs:
            todo += [s for s in e.free_symbols if s in self.defs]
        while todo:
            q = todo.pop()
            if q in need:
                continue
            need.add(q)
            todo += [s for s in self.defs[q].free_symbols if s in self.defs]
        return need


# --------------------------------------------------------------------------
# emitter
# --------------------------------------------------------------------------
class Emitter:
    def __init__(self, prob, plain=False):
        """plain: what a Maxima/gentran-generated pair looks like — no additive hints in the header, no factored
        tensor tables and no bp_derivsL_first() in the C file (the batched back-end then takes its general paths)"""
        self.p = prob
        self.plain = plain
        self.n = len(prob.x)
        self.m = len(prob.u)
        self._constraints()
        self.D = Deriver(prob)
        self.param_names = sorted(prob.params)  # internal consistency is all that matters
        self.time_syms = set(prob.x) | set(prob.u) | set(self.mu_c) | {self.w_pen}
        for nm in self.param_names:
            if prob.params[nm] == -1:
                self.time_syms.add(sp.Symbol("%s[k]" % nm, real=True))
        self._derive()
        self.cse = SharedTerms(self) if prob.cse else None
        self.tensor_tables = self._factor_tensors() if (prob.cse and not plain) else None

    def _factor_tensors(self):
        """{'basis': [products], 'xx': (coefficients, product number per slice), 'uu': ..., 'xu': ...} in the array
        order of fxx / fuu / fxu, if every entry of slice i (the second derivatives of f_i) is a number times ONE
        product shared by the whole slice (or zero); else None"""
        st = SharedTerms(self)
        basis, number, tables = [], {}, {}
        for nm, ten in (("xx", self.fxx), ("uu", self.fuu), ("xu", self.fxu)):
            items = self.jaco2_items("f" + nm, ten)
            per_slice = len(items) // self.n
            coefs, slices = [], []
            for i in range(self.n):
                shared = None
                for lhs, e in items[i * per_slice:(i + 1) * per_slice]:
                    terms = st.split(e) if e != 0 else []
                    if len(terms) > 1 or (terms and terms[0][1] is None):
                        return None
                    if not terms:
                        coefs.append(0.0)
                        continue
                    coef, prod = terms[0]
                    if shared is None:
                        shared = prod
                    elif prod != shared:
                        return None
                    coefs.append(float(coef))
                if shared is not None and shared not in number:
                    number[shared] = len(basis)
                    basis.append(shared)
                slices.append(number[shared] if shared is not None else 0)
            tables[nm] = (coefs, slices)
        if not basis:
            return None
        tables["basis"] = basis
        return tables

    # ---- augmented-Lagrangian terms (genenerator_main.mac:46-124) ----------
    def _constraints(self):
        """h and its penalty become auxiliaries (the reference assigns them with `::`), the penalty is added to
        L (running) or F (final); mu_<kind>_<i> print as m->mu_<kind>[i-1], w_pen as the local w_pen"""
        p = self.p
        self.w_pen = sp.Symbol("w_pen", real=True)
        self.mu_c = {}      # multiplier symbol -> C spelling
        self.cons_al = {}   # kind -> list of (h aux symbol, mu symbol)
        w = self.w_pen
        for kind, exprs in (("le", p.hle), ("li", p.hli), ("fe", p.hfe), ("fi", p.hfi)):
            self.cons_al[kind] = []
            for i, h in enumerate(exprs):
                h = sp.sympify(h)
                if kind[0] == "f" and any(uu in h.free_symbols for uu in p.u):
                    raise ValueError("h%s must not depend on any input u" % kind)  # genenerator_main.mac:48-49
                hs = p.auxiliary("h%s_%d" % (kind, i + 1), h)
                mu = sp.Symbol("mu_%s_%d" % (kind, i + 1), real=True)
                self.mu_c[mu] = "m->mu_%s[%d]" % (kind, i)
                if kind[1] == "e":
                    pen = mu * hs + sp.Float(0.5) * w * hs**2
                else:
                    pen = sp.Piecewise((mu * hs * (1 + w * hs), hs >= 0), (mu * hs / (1 - w * hs), True))
                ps = p.auxiliary("p%s_%d" % (kind, i + 1), pen)
                if kind[0] == "l":
                    p.L = p.L + ps
                else:
                    p.F = sp.sympify(p.F) + ps
                self.cons_al[kind].append((hs, mu))

    # ---- symbolic work -------------------------------------------------
    def _derive(self):
        p, D, n, m = self.p, self.D, self.n, self.m
        td = D.total_diff
        x, u = p.x, p.u
        self.fx = [[td(p.f[r], x[c]) for c in range(n)] for r in range(n)]
        self.fu = [[td(p.f[r], u[c]) for c in range(m)] for r in range(n)]
        self.fxx = [[[td(self.fx[r][c], x[d]) for d in range(n)] for c in range(n)] for r in range(n)]
        self.fuu = [[[td(self.fu[r][c], u[d]) for d in range(m)] for c in range(m)] for r in range(n)]
        self.fxu = [[[td(self.fx[r][c], u[d]) for d in range(m)] for c in range(n)] for r in range(n)]
        self.Lx = [td(p.L, x[r]) for r in range(n)]
        self.Lu = [td(p.L, u[r]) for r in range(m)]
        self.Lxx = [[td(self.Lx[r], x[c]) for c in range(n)] for r in range(n)]
        self.Luu = [[td(self.Lu[r], u[c]) for c in range(m)] for r in range(m)]
        self.Lxu = [[td(self.Lx[r], u[c]) for c in range(m)] for r in range(n)]
        if D.depends_on(p.F, set(u)):
            raise ValueError("F may not depend on u")  # genenerator_main.mac:127-128
        self.Fx = [td(p.F, x[r]) for r in range(n)]
        self.Fxx = [[td(self.Fx[r], x[c]) for c in range(n)] for r in range(n)]
        simp = (lambda e: e) if p.fast else (lambda e: _tidy(e) if e != 0 else e)
        for name in ("fx", "fu", "Lxx", "Luu", "Lxu", "Fxx"):
            setattr(self, name, [[simp(e) for e in row] for row in getattr(self, name)])
        for name in ("fxx", "fuu", "fxu"):
            setattr(self, name, [[[simp(e) for e in row] for row in mat] for mat in getattr(self, name)])
        for name in ("Lx", "Lu", "Fx"):
            setattr(self, name, [simp(e) for e in getattr(self, name)])

        # input constraints: each depends on exactly one input with coefficient +-1
        self.cons = []
        for i, h in enumerate(p.h):
            hu = [sp.simplify(sp.diff(h, uu)) for uu in u]
            nz = [j for j, c in enumerate(hu) if c != 0]
            if len(nz) != 1 or abs(hu[nz[0]]) != 1:
                raise ValueError("constraint %d must depend on one input with coefficient +-1" % (i + 1))
            j = nz[0]
            sign = int(hu[j])
            lim = sp.expand(h - sign * u[j])
            if sign > 0:
                lim = -lim
            hx = [td(h, xx) for xx in x]
            self.cons.append(dict(index=i, input=j, sign=sign, limit=lim, hx=hx, expr=h))
        self.has_hx = any(e != 0 for c in self.cons for e in c["hx"])

        # which dependent symbols are needed where
        first_order = list(itertools.chain(
            p.f, [p.L], *self.fx, *self.fu, self.Lx, self.Lu, *self.Lxx, *self.Luu, *self.Lxu,
            [c["limit"] for c in self.cons], *[c["hx"] for c in self.cons]))
        second_order = list(itertools.chain(
            *[itertools.chain(*mm) for mm in self.fxx],
            *[itertools.chain(*mm) for mm in self.fuu],
            *[itertools.chain(*mm) for mm in self.fxu]))
        self.run_need = D.closure(first_order)
        self.run_need_full = D.closure(second_order) - self.run_need
        self.fin_need = D.closure(list(itertools.chain([p.F], self.Fx, *self.Fxx)))
        for s in self.fin_need:
            if D.depends_on(D.defs[s], set(u)):
                raise ValueError("final-cost auxiliary %s depends on an input" % s)

    # ---- symbol -> C name ------------------------------------------------
    def csub(self, e, where):
        """substitute C spellings; where in {'run','fin'} selects struct prefix for aux"""
        e = sp.sympify(e)
        rep = {}
        for i, s in enumerate(self.p.x):
            rep[s] = sp.Symbol("x[%d]" % i)
        for i, s in enumerate(self.p.u):
            rep[s] = sp.Symbol("u[%d]" % i)
        for pi, nm in enumerate(self.param_names):
            sz = self.p.params[nm]
            if sz == 1:
                rep[sp.Symbol(nm, real=True)] = sp.Symbol("p[%d][0]" % pi)
            elif sz == -1:
                rep[sp.Symbol("%s[k]" % nm, real=True)] = sp.Symbol("p[%d][k]" % pi)
            else:
                for j in range(sz):
                    rep[sp.Symbol("%s[%d]" % (nm, j), real=True)] = sp.Symbol("p[%d][%d]" % (pi, j))
        for s in self.D.defs:
            rep[s] = sp.Symbol(self.macro_name(s))
        for mu, spelling in self.mu_c.items():
            rep[mu] = sp.Symbol(spelling)
        return e.xreplace(rep)

    def macro_name(self, s):
        return ("aux_" if self.D.kind[s] == "aux" else "daux_") + s.name

    def is_time_var(self, e):
        return self.D.depends_on(sp.sympify(e), self.time_syms)

    def is_const_number(self, e):
        return len(sp.sympify(e).free_symbols) == 0

    # ---- statement printers ---------------------------------------------
    def assign(self, lhs, e, ind=4, ret="0", guard=True, lhs_name=None):
        """lhs= e; followed by the NaN/Inf guard of genenerator_main.mac:189-199 unless e is a plain number
        (lhs_name: what the guard tests and prints when lhs is a declaration)"""
        pad = " " * ind
        e = sp.sympify(e)
        rhs = cexpr(self.csub(e, None))
        out = "%s%s= %s;\n" % (pad, lhs, rhs)
        if guard and not self.is_const_number(e):
            v = lhs_name or lhs
            out += ('%sif(isNANorINF(%s)) { PRNT("    @k %%d: %s in line %%d is nan or inf: %%g\\n", k, __LINE__-1, %s); return %s; }\n'
                    % (pad, v, v.replace('"', ""), v, ret))
        return out

    def block(self, items, want_time_var, ind=4, cse=None, pair=False):
        """items: list of (lhs, expr). emit those whose time-variance matches; cse: a SharedTerms that names
        the products the entries have in common.  pair: two neighbouring array entries are assigned first and
        guarded afterwards (same values, same guards; a back-end that writes the records to device memory gets one
        16-byte store instead of two of 8 bytes)"""
        todo = [(lhs, cse.rewrite(e) if cse else e) for lhs, e in items
                if want_time_var is None or self.is_time_var(e) == want_time_var]
        out, i = "", 0
        while i < len(todo):
            if pair and i + 1 < len(todo) and self._neighbours(todo[i][0], todo[i + 1][0]) and not self._ends_piece(todo[i][0]):
                a, b = self.assign(todo[i][0], todo[i][1], ind).split("\n")[:-1], self.assign(todo[i + 1][0], todo[i + 1][1], ind).split("\n")[:-1]
                # (each is the assignment, then its guard unless the value is a plain number)
                guards = [g.replace("__LINE__-1", "__LINE__-2") for g in a[1:] + b[1:]]
                out += "\n".join([a[0], b[0]] + guards) + "\n"
                i += 2
            else:
                out += self.assign(todo[i][0], todo[i][1], ind)
                i += 1
        return out

    def _ends_piece(self, lhs):
        """the entry is the last one of a piece of a run (see _record_runs): the next one is assigned behind its guard"""
        m = re.fullmatch(r"t->(\w+)\[(\d+)\]", lhs)
        place = self.record_offsets()
        return bool(m and m.group(1) in place and (place[m.group(1)] + int(m.group(2)) + 1) % self.RUN_MAX == 0)

    @staticmethod
    def _neighbours(l1, l2):
        m1, m2 = re.fullmatch(r"(.*)[\[(](\d+)[\])]", l1), re.fullmatch(r"(.*)[\[(](\d+)[\])]", l2)
        return bool(m1 and m2 and m1.group(1) == m2.group(1) and int(m2.group(2)) == int(m1.group(2)) + 1)

    def jaco_items(self, name, mat):
        nr, nc = len(mat), len(mat[0])
        return [("t->%s[%d]" % (name, r + c * nr), mat[r][c]) for c in range(nc) for r in range(nr)]

    def grad_items(self, name, v):
        return [("t->%s[%d]" % (name, r), v[r]) for r in range(len(v))]

    def hess_items(self, name, mat):
        nr, nc = len(mat), len(mat[0])
        items, idx = [], 0
        for c in range(nc):
            rows = range(c + 1) if nr == nc else range(nr)
            for r in rows:
                items.append(("t->%s[%d]" % (name, idx), mat[r][c]))
                idx += 1
        return items

    def jaco2_items(self, name, ten):
        n1, n2, n3 = len(ten), len(ten[0]), len(ten[0][0])
        items, idx = [], 0
        for r in range(n1):
            for d in range(n3):
                cols = range(d + 1) if n2 == n3 else range(n2)
                for c in cols:
                    items.append(("t->%s[%d]" % (name, idx), ten[r][c][d]))
                    idx += 1
        return items

    def all_zero(self, ten):
        return all(e == 0 for mat in ten for row in mat for e in row)

    # ---- aux ordering ----------------------------------------------------
    def aux_syms(self, need, kinds):
        return [s for s in self.D.order if s in need and self.D.kind[s] in kinds]

    def aux_block(self, need, kinds, want_time_var, u_dep=None, ind=4):
        out = ""
        for s in self.aux_syms(need, kinds):
            d = self.D.defs[s]
            tv = self.is_time_var(d)
            if want_time_var is not None and tv != want_time_var:
                continue
            if u_dep is not None and self.D.depends_on(d, set(self.p.u)) != u_dep:
                continue
            out += self.assign(self.macro_name(s), d, ind)
        return out

    # ---- files -------------------------------------------------------------
    def hints_block(self):
        if self.plain:
            return ""
        return f"""/* additive hints for the batched backend (absent in Maxima-generated headers,
 * which are then treated as the general case) */
#define ILQG_PROBLEM_NAME "{self.p.name}"
#define ILQG_STATE_DEPENDENT_LIMITS {1 if self.has_hx else 0}
#define ILQG_TENSOR_NBASIS {len(self.tensor_tables["basis"]) if self.tensor_tables else 0}  /* > 0: iLQG_func.c has the factored tensor tables */
#define ILQG_TENSOR_INIT_WRITES {1 if self.tensor_init_writes() else 0}  /* init_running() writes constant entries of fxx / fuu / fxu */
/* the derivative entries bp_derivsL() writes, X(member, index) each: all others are written once, by init_running() */
#define ILQG_TIME_VARYING(X) {self.time_varying_list(False)}
#if FULL_DDP
#define ILQG_TIME_VARYING_FULL(X) {self.time_varying_list(True)}
#else
#define ILQG_TIME_VARYING_FULL(X)
#endif
/* among the others: the entries that are identically 0 (what a dense back_pass multiplies by zero, matMult.c:3-72) */
#define ILQG_STRUCTURAL_ZERO(X) {self.time_varying_list(False, zeros=True)}
#if FULL_DDP
#define ILQG_STRUCTURAL_ZERO_FULL(X) {self.time_varying_list(True, zeros=True)}
#else
#define ILQG_STRUCTURAL_ZERO_FULL(X)
#endif

"""

    def problem_h(self):
        n, m = self.n, self.m
        run_members = "".join("    double %s;\n" % s.name for s in self.aux_syms(self.run_need, ("aux", "d1", "d2")))
        run_members_full = "".join("    double %s;\n" % s.name for s in self.aux_syms(self.run_need_full, ("aux", "d1", "d2")))
        fin_members = "".join("    double %s;\n" % s.name for s in self.aux_syms(self.fin_need, ("aux", "d1", "d2")))

        def mul_members(kinds):
            out = ""
            for kind in kinds:
                cnt = len(self.cons_al[kind])
                if cnt:
                    out += "    double mu_%s[%d];\n    double last_h%s[%d];\n" % (kind, cnt, kind, cnt)
            return out
        mul_el, mul_fin = mul_members(("le", "li")), mul_members(("fe", "fi"))
        return f"""/* Problem header for '{self.p.name}' emitted by tools/gen_problem.py. Do not edit.
 * Layout contract: reference iLQG_problem.tem:16-89. */
#ifndef ILQG_PROBLEM_H
#define ILQG_PROBLEM_H

#include <math.h>
#include "mex.h"
#ifndef  HAVE_OCTAVE
#include "matrix.h"
#endif

#define isNANorINF(v) (mxIsNaN(v) || mxIsInf(v))
#define INF mxGetInf()

#define N_X {n}
#define N_U {m}

#define sizeofQxx {n * (n + 1) // 2}
#define sizeofQuu {m * (m + 1) // 2}
#define sizeofQxu {n * m}

{self.hints_block()}typedef struct {{
    double x[N_X];
    double u[N_U];
    double lower[N_U];
    double upper[N_U];
    double lower_sign[N_U];
    double upper_sign[N_U];
    double lower_hx[N_X*N_U];
    double upper_hx[N_X*N_U];

    double l[N_U];
    double L[N_U*N_X];
    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
    double cu[N_U];
    double cuu[sizeofQuu];
    double cxu[sizeofQxu];
    double fx[N_X*N_X];
    double fu[N_X*N_U];
#if FULL_DDP
    double fxx[N_X*sizeofQxx];
    double fuu[N_X*sizeofQuu];
    double fxu[N_X*sizeofQxu];
#endif
{run_members}#if FULL_DDP
{run_members_full}#endif
}} trajEl_t;

typedef struct {{
    double x[N_X];

    double c;
    double cx[N_X];
    double cxx[sizeofQxx];
{fin_members}}} trajFin_t;

typedef struct {{
    trajEl_t* t;
    trajFin_t f;
}} traj_t;

typedef struct {{
{mul_el}}} multipliersEl_t;

typedef struct {{
{mul_fin}}} multipliersFin_t;

typedef struct {{
    multipliersEl_t* t;
    multipliersFin_t f;
}} multipliers_t;

#endif // ILQG_PROBLEM_H
"""

    # ---- iLQG_func.c ---------------------------------------------------------
    # Function names, signatures, what each function computes and in which order are the reference's contract
    # (iLQG_func.tem:40-521, the solver and the MEX shell call them by name); the bodies below are written for this
    # generator.  Inside every function the expressions printed by csub() refer to x[], u[], p[][], k, m-> and w_pen,
    # so each function that evaluates expressions declares those names first (prologue()).
    PROLOGUE = {
        "x": "const double *const x= t->x;",
        "u": "const double *const u= t->u;",
        "p": "double **const p= o->p;",
        "wl": "const double w_pen= o->w_pen_l;",
        "wf": "const double w_pen= o->w_pen_f;",
        "kN": "const int k= o->n_hor;",
    }

    def prologue(self, *names):
        return "".join("    %s\n" % self.PROLOGUE[n] for n in names) + "\n"

    def func_c(self):
        p, n, m = self.p, self.n, self.m
        o = []
        w = o.append
        w(f"/* Problem functions for '{p.name}' emitted by tools/gen_problem.py. Do not edit.\n"
          " * Function set, signatures and evaluation order: reference iLQG_func.tem:40-521. */\n")
        w('#include "iLQG.h"\n#include "matMult.h"\n\n')
        w("#define mcond(cond, a, dummy, b) ((cond)? a: b)\n#define sec(x) (1.0/cos(x))\n#define csc(x) (1.0/sin(x))\n\n")
        w("int n_params= %d;\n\n" % len(self.param_names))
        for i, nm in enumerate(self.param_names):
            w('tParamDesc p_name%d= {"%s", %d, 0};\n' % (i + 1, nm, p.params[nm]))
        w("int n_vars= 0;\n\n")
        w("tParamDesc *paramdesc[]= {%s};\n\n" % ", ".join("&p_name%d" % (i + 1) for i in range(len(self.param_names))))
        for s in self.D.order:
            if s in self.run_need or s in self.run_need_full or s in self.fin_need:
                w("#define %s t->%s\n" % (self.macro_name(s), s.name))
        w("\n")
        w("static int calcXVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o);\n"
          "static int calcXUVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o);\n"
          "static int calcFVariableAux(trajFin_t *t, multipliersFin_t *m, tOptSet *o);\n"
          "static int calcLAuxDeriv(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o);\n"
          "static int calcFAuxDeriv(trajFin_t *t, multipliersFin_t *m, tOptSet *o);\n"
          "static int bp_derivsL(trajEl_t *t, int k, double **p);\n"
          "static int bp_derivsF(trajFin_t *t, int k, double **p);\n\n")
        w(self.emit_cost_and_dynamics())
        w(self.emit_input_limits())
        w(self.emit_sweeps())
        w(self.emit_auxiliaries())
        w(self.emit_derivatives())
        w(self.emit_constants())
        w(self.emit_setup())
        w(self.multiplier_update())
        w("/* iLQG.c:236,337: multipliers of the running constraints, then of the final ones */\n"
          "int update_multipliers(tOptSet *o, int init) {\n"
          "    return update_multipliers_running(o, init) && update_multipliers_final(o, init);\n}\n\n"
          "/* no outputs g are defined by this generator (iLQG_func.tem:511-521) */\n"
          "int get_g_size() { return 0; }\n\n"
          "int calcG(double g[], trajEl_t *t, int k, double **p) { return 1; }\n")
        w(self.emit_factored_tensors())
        w(self.emit_step_parts())
        w(self.emit_deriv_parts())
        return "".join(o)

    def emit_cost_and_dynamics(self):
        p, n = self.p, self.n
        out = "/* running cost of one step, final cost, one step of the dynamics */\n"
        out += "static int ddpL(trajEl_t *t, int k, tOptSet *o) {\n" + self.prologue("x", "u", "p")
        out += self.assign("t->c", p.L) + "    return 1;\n}\n\n"
        out += "static int ddpF(trajFin_t *t, tOptSet *o) {\n" + self.prologue("x", "kN", "p")
        out += self.assign("t->c", p.F) + "    return 1;\n}\n\n"
        out += "static int ddpf(double x_next[], trajEl_t *t, int k, double **p, int N) {\n" + self.prologue("x", "u")
        for r in range(n):
            out += self.assign("x_next[%d]" % r, p.f[r])
        out += "    return 1;\n}\n\n"
        return out

    def emit_input_limits(self):
        """clampU: project u onto the input constraints, constraint by constraint in their given order;
        limitsU: the box those constraints leave for a CHANGE of u around the nominal input, and for each side of each
        input the constraint that is active there (sign of the input in it, its gradient with respect to x)."""
        n = self.n
        out = "void clampU(double *u, trajEl_t *t, int k, double **p, int N) {\n    const double *const x= t->x;\n    double bound;\n\n"
        for c in self.cons:
            j = c["input"]
            out += "    /* h[%d]= %s */\n" % (c["index"] + 1, sp.sstr(c["expr"]))
            out += self.assign("bound", c["limit"], guard=False)
            out += "    if(u[%d]%sbound) u[%d]= bound;\n" % (j, ">" if c["sign"] > 0 else "<", j)
        out += "}\n\n"

        out += ("static void limitsU(trajEl_t *t, int k, double **p, int N) {\n"
                "    const double *const x= t->x;\n"
                "    int active[2][N_U];  /* constraint that bounds input iu from below [0] / from above [1]; -1: none */\n"
                "    double bound;\n    int iu, side;\n\n"
                "    for(iu= 0; iu<N_U; iu++) {\n        active[0][iu]= active[1][iu]= -1;\n"
                "        t->lower[iu]= -INF;\n        t->upper[iu]= INF;\n    }\n\n")
        for c in self.cons:
            j, side = c["input"], (1 if c["sign"] > 0 else 0)
            arr, cmp_ = ("upper", ">") if side else ("lower", "<")
            out += "    /* h[%d]= %s */\n" % (c["index"] + 1, sp.sstr(c["expr"]))
            out += self.assign("bound", c["limit"], guard=False)
            out += "    if(t->%s[%d]%sbound) { t->%s[%d]= bound; active[%d][%d]= %d; }\n" % (arr, j, cmp_, arr, j, side, j, c["index"])
        out += ("\n    /* the solver works with the change of u */\n"
                "    for(iu= 0; iu<N_U; iu++) {\n        t->lower[iu]-= t->u[iu];\n        t->upper[iu]-= t->u[iu];\n    }\n\n"
                + ("" if self.plain else
                   "    /* additive: a back-end that will not read *_sign / *_hx of this element (limits that do not depend on the\n"
                   "     * state: constants) may say so through a condition of its own */\n"
                   "#ifndef ILQG_LIMIT_GRADIENTS_WANTED\n#define ILQG_LIMIT_GRADIENTS_WANTED 1\n#endif\n"
                   "    if(ILQG_LIMIT_GRADIENTS_WANTED)\n") +
                "    for(side= 0; side<2; side++) {\n"
                "        double *const sign= side? t->upper_sign: t->lower_sign;\n"
                "        double *const grad= side? t->upper_hx: t->lower_hx;\n"
                "        for(iu= 0; iu<N_U; iu++) {\n"
                "            double *const hx_= grad + iu*N_X;\n"
                "            switch(active[side][iu]) {\n")
        for c in self.cons:
            out += "                case %d:\n" % c["index"]
            for jx in range(n):
                out += self.assign("hx_[%d]" % jx, c["hx"][jx], ind=20, guard=False)
            out += "                    sign[iu]= %d.0;\n                    break;\n" % c["sign"]
        out += ("                default:  /* unbounded on this side: the gradient is not used */\n"
                "                    sign[iu]= 0.0;\n            }\n        }\n    }\n}\n\n")
        return out

    def emit_sweeps(self):
        return """/* Roll-out of candidate trajectory c (line_search.c:40, iLQG.c:338, iLQG_mex.c:116).
 * alpha != 0: u = u_nom + alpha*l + L (x - x_nom) with the gains of the nominal trajectory, accumulated state by
 * state; alpha == 0: the nominal inputs as they are.  cost_only: x and u of c are kept, only the cost is summed.
 * csum[0] holds the cost summed so far also when a NaN/Inf guard ends the sweep (return 0). */
int forward_pass(traj_t *c, tOptSet *o, double alpha, double *csum, int cost_only) {
    const int n_steps= o->n_hor;
    const int rollout= !cost_only;
    int k, ix, iu;

    csum[0]= 0.0;
    if(rollout)
        for(ix= 0; ix<N_X; ix++) c->t[0].x[ix]= o->x0[ix];

    for(k= 0; k<n_steps; k++) {
        const trajEl_t *const ref= o->nominal->t + k;
        trajEl_t *const cur= c->t + k;
        multipliersEl_t *const mul= o->multipliers.t + k;

        if(rollout) {
            if(alpha) {
                for(iu= 0; iu<N_U; iu++)
                    cur->u[iu]= ref->u[iu] + ref->l[iu]*alpha;
                for(ix= 0; ix<N_X; ix++) {
                    const double dev= cur->x[ix] - ref->x[ix];
                    for(iu= 0; iu<N_U; iu++)
                        cur->u[iu]+= ref->L[MAT_IDX(iu, ix, N_U)]*dev;
                }
            } else {
                for(iu= 0; iu<N_U; iu++)
                    cur->u[iu]= ref->u[iu];
            }
        }
        if(!calcXVariableAux(cur, mul, k, o)) return 0;
        if(rollout) clampU(cur->u, cur, k, o->p, n_steps);
        if(!calcXUVariableAux(cur, mul, k, o)) return 0;
        if(rollout && !ddpf((k+1<n_steps)? c->t[k+1].x: c->f.x, cur, k, o->p, n_steps)) return 0;
        if(!ddpL(cur, k, o)) return 0;
        csum[0]+= cur->c;
    }

    if(!calcFVariableAux(&c->f, &o->multipliers.f, o)) return 0;
    if(!ddpF(&c->f, o)) return 0;
    csum[0]+= c->f.c;
    return 1;
}

/* Derivatives along the nominal trajectory (iLQG.c:247): the final step, then the running steps from the end of
 * the horizon to its start, each with the box its input constraints leave around the nominal input. */
int calc_derivs(tOptSet *o) {
    const int n_steps= o->n_hor;
    traj_t *const nom= o->nominal;
    int k;

    if(!calcFAuxDeriv(&nom->f, &o->multipliers.f, o)) return 0;
    if(!bp_derivsF(&nom->f, n_steps, o->p)) return 0;

    for(k= n_steps; k-->0; ) {
        trajEl_t *const el= nom->t + k;
        if(!calcLAuxDeriv(el, o->multipliers.t + k, k, o)) return 0;
        if(!bp_derivsL(el, k, o->p)) return 0;
        limitsU(el, k, o->p, n_steps);
    }
    return 1;
}

"""

    def emit_auxiliaries(self):
        out = "/* auxiliary variables: members of the step's element, evaluated once and reused by everything that follows */\n"
        out += "static int calcXVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {\n" + self.prologue("x", "p", "wl")
        out += self.aux_block(self.run_need | self.run_need_full, ("aux",), True, u_dep=False) + "    return 1;\n}\n\n"
        out += "static int calcXUVariableAux(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {\n" + self.prologue("x", "u", "p", "wl")
        out += self.aux_block(self.run_need | self.run_need_full, ("aux",), True, u_dep=True) + "    return 1;\n}\n\n"
        out += "static int calcFVariableAux(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {\n" + self.prologue("x", "p", "wf", "kN")
        out += self.aux_block(self.fin_need, ("aux",), True) + "    return 1;\n}\n\n"
        out += "static int calcLAuxDeriv(trajEl_t *t, multipliersEl_t *m, int k, tOptSet *o) {\n" + self.prologue("x", "u", "wl", "p")
        out += self.aux_block(self.run_need, ("d1", "d2"), True)
        out += "#if FULL_DDP\n" + self.aux_block(self.run_need_full, ("d1", "d2"), True) + "#endif\n    return 1;\n}\n\n"
        return out

    def emit_derivatives(self):
        """time-varying entries of the derivative record of one step / of the final step"""
        cse = self.cse
        out = "static int bp_derivsL(trajEl_t *t, int k, double **p) {\n    const double *const x= t->x;\n    const double *const u= t->u;\n\n"
        if cse:
            cse.use("first")
        pair = bool(self.tensor_tables)  # the function batched back-ends call per step (bp_derivsL_first)
        first = (self.block(self.jaco_items("fx", self.fx), True, cse=cse, pair=pair) + "\n" +
                 self.block(self.jaco_items("fu", self.fu), True, cse=cse, pair=pair) + "\n")
        cost = ""
        for items in (self.grad_items("cx", self.Lx), self.hess_items("cxx", self.Lxx), self.grad_items("cu", self.Lu),
                      self.hess_items("cuu", self.Luu), self.hess_items("cxu", self.Lxu)):
            cost += self.block(items, True, cse=cse, pair=pair) + "\n"
        cse2 = cse
        if self.tensor_tables:
            cse2 = SharedTerms(self, "ct")  # a function of its own (bp_derivsL_second): its own names
        if cse2:
            cse2.use("second")
        second = ""
        for nm, ten in (("fxx", self.fxx), ("fuu", self.fuu), ("fxu", self.fxu)):
            if not self.all_zero(ten):
                second += self.block(self.jaco2_items(nm, ten), True, cse=cse2) + "\n"
        if cse2:
            second = "    /* products shared by the entries of the tensors */\n" + cse2.declarations("second") + "\n" + second
            if not self.tensor_tables:
                out += "    /* products shared by several entries */\n" + cse.declarations("first") + "\n"
        if self.tensor_tables:
            # the part a back-end that evaluates the tensors from the factored tables still needs, on its own
            head = "    const double *const x= t->x;\n    const double *const u= t->u;\n\n"
            self.direct_entries = []
            first, cost = self._record_runs(first), self._record_runs(cost)  # (fx and fu are one block: `first`)
            direct = ("/* the entries bp_derivsL_first assigns outside the runs, as X(member, index) ... */\n#define ILQG_REC_DIRECT(X) " +
                      " ".join("X(%s, %d)" % e for e in self.direct_entries) + "\n")
            taken, body = self._aux_once("    /* products shared by several entries */\n" + cse.declarations("first") + "\n"
                                         "    /* dynamics */\n" + first + "    /* cost */\n" + cost)
            out = (self.RECORD_MACROS + direct + "static int bp_derivsL_first(trajEl_t *t, int k, double **p) {\n" + head + taken + body +
                   "    return 1;\n}\n\n"
                   "#if FULL_DDP\nstatic int bp_derivsL_second(trajEl_t *t, int k, double **p) {\n" + head +
                   second + "    return 1;\n}\n#endif\n\n"
                   "static int bp_derivsL(trajEl_t *t, int k, double **p) {\n    if(!bp_derivsL_first(t, k, p)) return 0;\n"
                   "#if FULL_DDP\n    if(!bp_derivsL_second(t, k, p)) return 0;\n#endif\n    return 1;\n}\n\n")
        else:
            out += "    /* dynamics */\n" + first + "#if FULL_DDP\n" + second + "#endif\n    /* cost */\n" + cost + "    return 1;\n}\n\n"

        out += "static int calcFAuxDeriv(trajFin_t *t, multipliersFin_t *m, tOptSet *o) {\n" + self.prologue("x", "wf", "p", "kN")
        out += self.aux_block(self.fin_need, ("d1", "d2"), True) + "    return 1;\n}\n\n"
        out += "static int bp_derivsF(trajFin_t *t, int k, double **p) {\n    const double *const x= t->x;\n\n"
        out += self.block(self.grad_items("cx", self.Fx), True) + "\n" + self.block(self.hess_items("cxx", self.Fxx), True)
        out += "    return 1;\n}\n\n"
        return out

    # Runs of neighbouring record entries that bp_derivsL_first assigns one after the other (all of fx, fu, cx, cu of a
    # typical problem) are written through ILQG_REC(member, index), each run closed by ILQG_REC_DONE(member, first, count)
    # in pieces of at most RUN_MAX entries.  By default that is the plain assignment t->member[index]= ...; a batched
    # back-end whose lanes each own a record may collect a run on chip and store it as whole cache lines.
    RUN_MIN, RUN_MAX = 8, 64
    RECORD_MACROS = ("#ifndef ILQG_REC  /* a back-end may define these two before including this file */\n"
                     "#define ILQG_REC(member, index) t->member[index]\n"
                     "#define ILQG_REC_DONE(member, first, count)  /* entries first .. first+count-1 have been assigned */\n"
                     "#endif\n")

    @staticmethod
    def _aux_once(text, ind=4):
        """(declarations, text): every auxiliary the statements read, taken into a local once at the head of the function
        (a back-end whose elements live in device memory gets all these loads in flight together instead of one round
        trip in front of each use; the values and the arithmetic are the same)"""
        names = []
        for m in re.finditer(r"\b(d?aux_\w+)\b", text):
            if m.group(1) not in names:
                names.append(m.group(1))
        if not names:
            return "", text
        decl = " " * ind + "/* auxiliaries read here, taken once */\n"
        for i in range(0, len(names), 4):
            decl += " " * ind + "const double " + ", ".join("v_%s= %s" % (n, n) for n in names[i:i + 4]) + ";\n"
        text = re.sub(r"\b(d?aux_\w+)\b", lambda m: "v_" + m.group(1), text)
        # ... and their sines and cosines, ahead of the first guard: one evaluation per auxiliary and function, in a block
        # that is always executed (two functions inlined into one caller then share the evaluations)
        trig = []
        for m in re.finditer(r"\b(sin|cos)\((v_d?aux_\w+)\)", text):
            if (m.group(1), m.group(2)) not in trig:
                trig.append((m.group(1), m.group(2)))
        if trig:
            decl += " " * ind + "/* ... and their sines and cosines */\n"
            for i in range(0, len(trig), 4):
                decl += " " * ind + "const double " + ", ".join("%s_%s= %s(%s)" % (f, a, f, a) for f, a in trig[i:i + 4]) + ";\n"
            text = re.sub(r"\b(sin|cos)\((v_d?aux_\w+)\)", lambda m: "%s_%s" % (m.group(1), m.group(2)), text)
        return decl + "\n", text

    def record_offsets(self):
        """place (in doubles from the start of trajEl_t) of the members bp_derivsL_first assigns: the layout of
        problem_h() up to fu, the same with and without FULL_DDP"""
        n, m = self.n, self.m
        off, out = 0, {}
        for name, size in (("x", n), ("u", m), ("lower", m), ("upper", m), ("lower_sign", m), ("upper_sign", m), ("lower_hx", n * m),
                           ("upper_hx", n * m), ("l", m), ("L", m * n), ("c", 1), ("cx", n), ("cxx", n * (n + 1) // 2), ("cu", m),
                           ("cuu", m * (m + 1) // 2), ("cxu", n * m), ("fx", n * n), ("fu", n * m)):
            out[name] = off
            off += size
        return out

    def _record_runs(self, text):
        """A run: entries assigned one after the other that are neighbours in the ELEMENT (fx[255] and fu[0] are), at
        least RUN_MIN of them.  It is closed in pieces that end where the place in the element is a multiple of RUN_MAX
        (so a back-end that writes a piece as whole cache lines meets every line once), each piece named by its first
        entry: ILQG_REC_DONE(member, index, count) — the count may reach into the next member."""
        lines = text.split("\n")
        place = self.record_offsets()
        seq = []  # (member, index) in the order of their assignments
        for ln in lines:
            m = re.match(r"\s*t->(\w+)\[(\d+)\]= ", ln)
            if m:
                seq.append((m.group(1), int(m.group(2))))
        at = lambda e: place[e[0]] + e[1]
        pieces, i = [], 0  # (entries of the piece)
        while i < len(seq):
            j = i
            while j + 1 < len(seq) and seq[j + 1][0] in place and seq[j][0] in place and at(seq[j + 1]) == at(seq[j]) + 1:
                j += 1
            if j - i + 1 >= self.RUN_MIN and seq[i][0] in place:
                a = i
                while a <= j:
                    e = a
                    while e < j and (at(seq[e]) + 1) % self.RUN_MAX != 0:
                        e += 1
                    pieces.append(seq[a:e + 1])
                    a = e + 1
            i = j + 1
        staged = {e: pc for pc in map(tuple, pieces) for e in pc}
        self.direct_entries = getattr(self, "direct_entries", []) + [e for e in seq if e not in staged]
        out, last_line = [], {}
        for n_, ln in enumerate(lines):
            def sub(mo):
                key = (mo.group(1), int(mo.group(2)))
                if key in staged:
                    last_line[staged[key]] = n_
                    return "ILQG_REC(%s, %d)" % key
                return mo.group(0)
            # (not inside the guard's message: it names the entry as the reference's files do)
            parts = re.split(r'("(?:[^"\\]|\\.)*")', ln)
            out.append("".join(q if q.startswith('"') else re.sub(r"t->(\w+)\[(\d+)\]", sub, q) for q in parts))
        done = {}
        for pc, n_ in last_line.items():
            done.setdefault(n_, []).append(pc)
        res = []
        for n_, ln in enumerate(out):
            res.append(ln)
            for pc in sorted(done.get(n_, []), key=lambda r: at(r[0])):
                res.append("    ILQG_REC_DONE(%s, %d, %d)" % (pc[0][0], pc[0][1], len(pc)))
        return "\n".join(res)

    def time_varying_list(self, full, zeros=False):
        """the entries emit_derivatives() assigns in bp_derivsL, as X(member, index) ...; zeros: instead, the entries
        whose expression is identically 0"""
        if full:
            groups = [self.jaco2_items(nm, ten) for nm, ten in (("fxx", self.fxx), ("fuu", self.fuu), ("fxu", self.fxu))
                      if zeros or not self.all_zero(ten)]
        else:
            groups = [self.jaco_items("fx", self.fx), self.jaco_items("fu", self.fu), self.grad_items("cx", self.Lx),
                      self.hess_items("cxx", self.Lxx), self.grad_items("cu", self.Lu), self.hess_items("cuu", self.Luu),
                      self.hess_items("cxu", self.Lxu)]
        out = []
        for items in groups:
            for lhs, e in items:
                if (sp.sympify(e) == 0) if zeros else self.is_time_var(e):
                    m = re.fullmatch(r"t->(\w+)\[(\d+)\]", lhs)
                    out.append("X(%s, %s)" % (m.group(1), m.group(2)))
        return " ".join(out)

    def tensor_init_writes(self):
        """whether init_running() stores anything in the second-derivative tensors of the dynamics (a back-end that
        works from the factored tables never reads them and may then leave them out of its records)"""
        return any(self.all_zero(ten) or self.block(self.jaco2_items(nm, ten), False, ind=8).strip() != ""
                   for nm, ten in (("fxx", self.fxx), ("fuu", self.fuu), ("fxu", self.fxu)))

    def emit_constants(self):
        """entries that do not change along a trajectory are written once (init_opt), element by element"""
        out = ("/* constant entries of every element of a trajectory buffer */\n"
               "static int init_running(trajEl_t *t, tOptSet *o) {\n    double **const p= o->p;\n    trajEl_t *const end= t + o->n_hor;\n    int k= 0;\n\n"
               "    for(; t<end; t++, k++) {\n")
        out += self.aux_block(self.run_need | self.run_need_full, ("aux",), False, ind=8)
        out += self.aux_block(self.run_need, ("d1", "d2"), False, ind=8)
        out += "#if FULL_DDP\n" + self.aux_block(self.run_need_full, ("d1", "d2"), False, ind=8) + "#endif\n"
        out += "        /* cost */\n"
        for items in (self.grad_items("cx", self.Lx), self.hess_items("cxx", self.Lxx), self.grad_items("cu", self.Lu),
                      self.hess_items("cuu", self.Luu), self.hess_items("cxu", self.Lxu)):
            out += self.block(items, False, ind=8) + "\n"
        out += "        /* dynamics */\n"
        out += self.block(self.jaco_items("fx", self.fx), False, ind=8) + "\n"
        out += self.block(self.jaco_items("fu", self.fu), False, ind=8) + "\n#if FULL_DDP\n"
        for nm, ten, sz in (("fxx", self.fxx, "N_X*sizeofQxx"), ("fuu", self.fuu, "N_X*sizeofQuu"), ("fxu", self.fxu, "N_X*sizeofQxu")):
            if self.all_zero(ten):
                out += "        { int e_; for(e_= 0; e_<%s; e_++) t->%s[e_]= 0.0; }\n" % (sz, nm)
            else:
                out += self.block(self.jaco2_items(nm, ten), False, ind=8)
            out += "\n"
        out += "#endif\n    }\n    return 1;\n}\n\n"
        out += "static int init_final(trajFin_t *t, tOptSet *o) {\n" + self.prologue("p", "kN")
        out += self.aux_block(self.fin_need, ("aux",), False) + self.aux_block(self.fin_need, ("d1", "d2"), False)
        out += self.block(self.grad_items("cx", self.Fx), False) + "\n" + self.block(self.hess_items("cxx", self.Fxx), False)
        out += "    return 1;\n}\n\n"
        return out

    def emit_setup(self):
        return ("int init_trajectory(traj_t *t, tOptSet *o) {\n    return init_running(t->t, o) && init_final(&t->f, o);\n}\n\n"
                + self.multiplier_init() +
                "int init_multipliers(tOptSet *o) {\n    return init_multipliers_running(o) && init_multipliers_final(o);\n}\n\n"
                "/* iLQG_mex.c:108: constants of every trajectory buffer; buffer 0 starts as the nominal trajectory, the\n"
                " * others as line-search candidates; multipliers at their start values */\n"
                "int init_opt(tOptSet *o) {\n    int b;\n\n"
                "    for(b= 0; b<=NUMBER_OF_THREADS; b++) {\n"
                "        if(!init_trajectory(&o->trajectories[b], o)) return 0;\n"
                "        if(b==0) o->nominal= &o->trajectories[0];\n"
                "        else o->candidates[b-1]= &o->trajectories[b];\n    }\n"
                "    return init_multipliers(o);\n}\n\n")


def _multiplier_init(self):
    """start values of the multipliers (iLQG_func.tem:371-393): equality constraints 0, inequality constraints 1,
    no constraint value remembered yet"""
    al = self.cons_al

    def fill(kinds, ind):
        pad, out = " " * ind, ""
        for kind in kinds:
            if al[kind]:
                out += ("%sfor(i= 0; i<%d; i++) { m->mu_%s[i]= %s; m->last_h%s[i]= 0.0; }\n"
                        % (pad, len(al[kind]), kind, "0.0" if kind[1] == "e" else "1.0", kind))
        return out

    out = "static int init_multipliers_running(tOptSet *o) {\n"
    if al["le"] or al["li"]:
        out += ("    multipliersEl_t *m= o->multipliers.t;\n    multipliersEl_t *const end= m + o->n_hor;\n    int i;\n\n"
                "    for(; m<end; m++) {\n" + fill(("le", "li"), 8) + "    }\n")
    out += "    return 1;\n}\n\nstatic int init_multipliers_final(tOptSet *o) {\n"
    if al["fe"] or al["fi"]:
        out += "    multipliersFin_t *const m= &o->multipliers.f;\n    int i;\n\n" + fill(("fe", "fi"), 4)
    out += "    return 1;\n}\n\n"
    return out


def _multiplier_update(self):
    """update_multipliers_running / _final: same effect as iLQG_func.tem:419-509.  The constraint values are the
    auxiliaries the last roll-out of the nominal trajectory left in its elements.  Per constraint: does its violation
    stall (larger than the tolerance and not reduced by the factor w_pen_fact1 since the last call)?  Then the
    violation is remembered and — unless this is the solver-entry call (init) — the multiplier is updated;
    equality: mu += w_pen*h, inequality after Ruxton.  A stalled violation raises the penalty weight once per call.
    With init != 0 the running part stops after the first element (as the reference's template does: its
    `if(init) return 1;` sits inside the loop over the horizon)."""
    al, w = self.cons_al, self.w_pen

    def hval(hs):
        return self.macro_name(hs)

    def progress(kind, ind):
        pad, out = " " * ind, ""
        for i, (hs, mu) in enumerate(al[kind]):
            h = hval(hs)
            if kind[1] == "e":
                out += "%sstalled|= violation_stalls(fabs(%s), fabs(m->last_h%s[%d]), o);\n" % (pad, h, kind, i)
            else:
                out += "%sstalled|= violation_stalls(%s, m->last_h%s[%d], o);\n" % (pad, h, kind, i)
            out += "%sm->last_h%s[%d]= %s;\n" % (pad, kind, i, h)
        return out

    def steps(kind, ind):
        pad, out = " " * ind, ""
        for i, (hs, mu) in enumerate(al[kind]):
            lhs = self.mu_c[mu]
            if kind[1] == "e":
                out += self.assign(lhs, mu + w * hs, ind)
            else:  # D. Ruxton's update for inequality constraints: active / inactive branch
                out += "%sif(%s>=0) {\n" % (pad, hval(hs))
                out += self.assign(lhs, mu * (1 + 2 * w * hs), ind + 4)
                out += "%s} else {\n" % pad
                out += self.assign(lhs, mu * (1 - w * hs) ** -2, ind + 4)
                out += "%s}\n" % pad
        return out

    any_al = any(al[k] for k in al)
    out = ""
    if any_al:
        out += ("/* a violation v stalls: above the tolerance and not smaller than 1/w_pen_fact1 of the one remembered */\n"
                "static int violation_stalls(double v, double last, const tOptSet *o) {\n"
                "    return v>o->tolConstraint && o->w_pen_fact1*v>last;\n}\n\n")
    out += "static int update_multipliers_running(tOptSet *o, int init) {\n"
    if al["le"] or al["li"]:
        out += ("    trajEl_t *t= o->nominal->t;\n    multipliersEl_t *m= o->multipliers.t;\n"
                "    const double w_pen= o->w_pen_l;\n    double **const p= o->p;\n    int stalled= 0, k;\n\n"
                "    for(k= 0; k<o->n_hor; k++, t++, m++) {\n")
        out += progress("le", 8) + progress("li", 8)
        out += "        if(init) return 1;  /* solver entry: the violations of the first element are remembered, nothing else */\n"
        out += steps("le", 8) + steps("li", 8)
        out += ("    }\n    if(!init && stalled)\n"
                "        o->w_pen_l= min(o->w_pen_max_l, o->w_pen_l*o->w_pen_fact1);\n")
    out += "    return 1;\n}\n\nstatic int update_multipliers_final(tOptSet *o, int init) {\n"
    if al["fe"] or al["fi"]:
        out += ("    trajFin_t *const t= &o->nominal->f;\n    multipliersFin_t *const m= &o->multipliers.f;\n"
                "    const double w_pen= o->w_pen_f;\n    double **const p= o->p;\n    const int k= o->n_hor;\n    int stalled= 0;\n\n")
        out += progress("fe", 4) + progress("fi", 4)
        out += ("    if(!init && stalled)\n        o->w_pen_f= min(o->w_pen_max_f, o->w_pen_f*o->w_pen_fact1);\n"
                "    if(init) return 1;\n")
        out += steps("fe", 4) + steps("fi", 4)
    out += "    return 1;\n}\n\n"
    return out


Emitter.multiplier_init = _multiplier_init
Emitter.multiplier_update = _multiplier_update


# --------------------------------------------------------------------------
# common factors of derivative entries
# --------------------------------------------------------------------------
class SharedTerms:
    """Splits derivative entries into sums of (number) * (product of everything else) and gives every distinct
    product a local name, so that a function evaluates it once however many entries contain it — the large tensors
    of a problem whose nonlinearity enters through a few auxiliaries are thousands of numeric multiples of a few dozen
    products.  (The reference leaves common subexpressions to gentran's optimiser.)  Two pools: products needed by
    first-order entries are declared unconditionally, products only the second derivatives of f need inside
    #if FULL_DDP (their auxiliaries only exist there)."""

    def __init__(self, emitter, prefix="cs"):
        self.em = emitter
        self.prefix = prefix
        self.pools = {"first": [], "second": []}
        self.names = {}
        self.pool = "first"

    def use(self, pool):
        self.pool = pool

    def split(self, e):
        """[(coefficient, product or None)] with e == sum(coefficient * product)"""
        out = []
        for term in sp.Add.make_args(sp.sympify(e)):
            coef, rest = term.as_coeff_Mul()
            out.append((coef, None if rest == 1 else rest))
        return out

    def name_of(self, product):
        if product not in self.names:
            self.names[product] = sp.Symbol("%s%d" % (self.prefix, len(self.names)))
            self.pools[self.pool].append(product)
        return self.names[product]

    def rewrite(self, e):
        return sp.Add(*[coef if prod is None else coef * self.name_of(prod) for coef, prod in self.split(e)])

    def declarations(self, pool, ind=4):
        out = ""
        for prod in self.pools[pool]:
            out += self.em.assign("const double %s" % self.names[prod].name, prod, ind, lhs_name=self.names[prod].name)
        return out


def _emit_factored_tensors(self):
    """Additive (no counterpart in the reference): when every entry of fxx, fuu, fxu is one number times one shared
    product, the same facts as tables, for back-ends that evaluate the tensors instead of storing them."""
    if not self.tensor_tables:
        return ""
    T = self.tensor_tables
    out = ("\n#if FULL_DDP\n/* ---- additive: the second derivatives of the dynamics in factored form (batched back-ends; the\n"
           " * reference's solver never reads this).  Every entry of slice i (the second derivatives of f_i) is a number\n"
           " * times ONE product shared by the slice,\n"
           " *     t->fxx[i*sizeofQxx + e] == ilqg_tensor_coef_xx[i*sizeofQxx + e] * basis[ilqg_tensor_slice_xx[i]]   (likewise fuu, fxu)\n"
           " * and bp_tensor_basis() evaluates the ILQG_TENSOR_NBASIS products of one step exactly as bp_derivsL does. */\n"
           "#ifndef ILQG_BASIS  /* a back-end may define these two before including this file */\n"
           "#define ILQG_BASIS(index) basis[index]\n"
           "#define ILQG_BASIS_DONE(count)  /* all products have been assigned */\n#endif\n"
           "static int bp_tensor_basis(double *basis, trajEl_t *t, int k, double **p) {\n"
           "    const double *const x= t->x;\n    const double *const u= t->u;\n\n")
    taken, body = self._aux_once(self.block([("ILQG_BASIS(%d)" % i, prod) for i, prod in enumerate(T["basis"])], None, pair=True))
    out += taken + body + "    ILQG_BASIS_DONE(%d)\n    return 1;\n}\n\n" % len(T["basis"])
    for nm in ("xx", "uu", "xu"):
        coef, slices = T[nm]
        out += "static const double ilqg_tensor_coef_%s[%d]= {\n" % (nm, len(coef))
        for i in range(0, len(coef), 6):
            out += "    " + ", ".join(cexpr(sp.Float(c)) if c != 0 else "0.0" for c in coef[i:i + 6]) + ",\n"
        out += "};\nstatic const int ilqg_tensor_slice_%s[N_X]= {%s};\n" % (nm, ", ".join(str(b) for b in slices))
    out += "#endif\n"
    return out


Emitter.emit_factored_tensors = _emit_factored_tensors


def _emit_step_parts(self):
    """Additive (no counterpart in the reference): one step of forward_pass cut into N_X independent parts, for
    back-ends that evaluate a trajectory's step on several wavefronts at once.  Part r owns component r of the dynamics
    and every N_X-th summand of the running cost, with the auxiliaries they need as locals (no trajEl_t: nothing but x
    and u goes in).  The assignments are the ones of calcXVariableAux / calcXUVariableAux / ddpf, unchanged; the cost is
    emitted summand by summand in the order ddpL's expression adds them, so that
        c = term[0]; c = c + term[1]; ...   reproduces ddpL's t->c to the last bit
    (a sum a + b + c is evaluated left to right; a printed "- q" is the addition of the negated summand).  NaN / Inf in a
    guarded value sets bad[0] instead of returning (the caller decides per trajectory).  Only for problems without
    multipliers (their auxiliaries take the multipliers and penalty weights as inputs)."""
    if self.plain or self.mu_c or not self.p.f:
        return ""
    # a back-end hands clampU() nothing but x (no element with evaluated auxiliaries): limits written with an auxiliary
    # would read beyond it
    if any(sp.sympify(c["limit"]).free_symbols & set(self.D.defs) for c in self.cons):
        return ""
    n = self.n
    L = self.csub(self.p.L, None)
    terms = _printer._as_ordered_terms(L, order=None) if L.is_Add else [L]
    # the pieces must print exactly as the whole does: check the reconstruction against the printer's own output
    whole = cexpr(L)
    rebuilt = ""
    for i, t in enumerate(terms):
        ts = cexpr(t)
        if i == 0:
            rebuilt = ts
        elif ts.startswith("-"):
            rebuilt += " - " + ts[1:]
        else:
            rebuilt += " + " + ts
    if rebuilt != whole:
        return ""  # (an expression shape the printer arranges differently: no parts for this problem)
    need_all = set()
    parts = []
    for r in range(n):
        mine_terms = [m for m in range(len(terms)) if m % n == r]
        exprs = [self.p.f[r]]
        # the summands are in C spelling already: find the auxiliaries they use through the macro names
        aux_by_macro = {self.macro_name(q): q for q in self.D.defs}
        used = set()
        for m in mine_terms:
            for sym in terms[m].free_symbols:
                if sym.name in aux_by_macro:
                    used.add(aux_by_macro[sym.name])
        need = self.D.closure(exprs) | used | self.D.closure([self.D.defs[q] for q in used])
        need = {q for q in need if self.D.kind[q] == "aux"}
        need_all |= need
        parts.append((r, mine_terms, need))
    members = [q for q in self.D.order if q in need_all]

    import re as _re

    def part_math(text):  # sin / cos of the parts go through macros a back-end may point at an inlined implementation
        text = _re.sub(r"(?<![A-Za-z0-9_])sin\(", "ILQG_PART_SIN(", text)
        return _re.sub(r"(?<![A-Za-z0-9_])cos\(", "ILQG_PART_COS(", text)

    def guarded(lhs, rhs):
        # (not `v - v == 0`: where v is a product, FMA contraction turns v - v into the product's rounding error)
        return "        %s= %s;\n        if(!(fabs(%s) <= 1.7976931348623157e308)) bad[0]= 1;\n" % (lhs, part_math(rhs), lhs)

    out = ("\n/* ---- additive: one step of forward_pass in ILQG_ROLLOUT_PARTS independent parts (batched back-ends that put\n"
           " * several wavefronts on a trajectory's step; the reference's solver never calls this).  Part r: component r of the\n"
           " * dynamics and the summands r, r + N_X, ... of the running cost, term[] indexed by their place in ddpL's sum:\n"
           " * t->c == ((term[0] + term[1]) + term[2]) + ...  A NaN or Inf in a guarded value sets bad[0]. */\n"
           "#define ILQG_ROLLOUT_PARTS %d\n#define ILQG_ROLLOUT_TERMS %d\n"
           "#ifndef ILQG_PART_SIN  /* a back-end may define these two before including this file */\n"
           "#define ILQG_PART_SIN(v) sin(v)\n#define ILQG_PART_COS(v) cos(v)\n#endif\n"
           "#ifndef ILQG_PART_FN  /* ... and the function's storage class / attributes */\n#define ILQG_PART_FN static\n#endif\n" % (n, len(terms)))
    out += "typedef struct {\n" + "".join("    double %s;\n" % q.name for q in members) + ("    double unused_;\n" if not members else "") + "} ilqg_step_aux_t;\n"
    out += ("ILQG_PART_FN void ilqg_step_part(int part, double x_next[], double term[], int bad[], const double *x, const double *u, int k, double **p, int N) {\n"
            "    ilqg_step_aux_t aux_, *const t= &aux_;\n\n    switch(part) {\n")
    for r, mine_terms, need in parts:
        out += "    case %d:\n" % r
        for q in self.D.order:
            if q in need:
                out += guarded(self.macro_name(q), cexpr(self.csub(self.D.defs[q], None)))
        out += guarded("x_next[%d]" % r, cexpr(self.csub(self.p.f[r], None)))
        for m in mine_terms:
            ts = cexpr(terms[m])
            out += "        term[%d]= %s;\n" % (m, part_math(ts))
        out += "        break;\n"
    out += "    default: break;\n    }\n}\n"
    return out


Emitter.emit_step_parts = _emit_step_parts


def _emit_deriv_parts(self):
    """Additive (no counterpart in the reference): the time-varying part of a step's derivative record in PARTS of
    ILQG_DERIV_PART_OUT outputs, for back-ends that assemble records in on-chip memory and write them as whole cache lines
    (one lane per (trajectory, step) storing into its own struct touches 64 lines per store instruction).  Only for problems
    with the factored tensor tables, without multipliers, without auxiliary DERIVATIVES among the time-varying members and
    with limits that do not depend on the state.

      ilqg_deriv_prepare()  the time-varying auxiliaries of the step (the assignments of calcXVariableAux /
                            calcXUVariableAux, unchanged), sin / cos of every auxiliary the derivatives take them of — ONCE
                            (bp_derivsL_first and bp_tensor_basis each evaluate them) —, and from them the products the
                            entries share (prod[]: the `cs` locals of bp_derivsL_first) and the products of
                            bp_tensor_basis (basis[]);
      ilqg_deriv_part(q)    outputs [q * OUT, (q + 1) * OUT) of the list ILQG_DERIV_OUTPUTS: the record's fxx[0 .. NBASIS)
                            (the products), the box limitsU() leaves for a change of u, the entries bp_derivsL_first writes —
                            each by the expression of the function it comes from, printed by the same printer (only
                            `sin(aux_..)` / `cos(aux_..)` are replaced, in the printed text, by the values prepared), so
                            the FMA-free build gives the same bits.
    A NaN or Inf in a guarded value sets bad[0]."""
    if self.plain or self.mu_c or not self.tensor_tables or self.has_hx or not self.p.f:
        return ""
    D = self.D
    tv_aux = [q for q in D.order if q in (self.run_need | self.run_need_full) and self.is_time_var(D.defs[q])]
    if any(D.kind[q] != "aux" for q in tv_aux):
        return ""
    cse = self.cse
    OUT = 16
    import re as _re
    trig_args = []  # auxiliary macro names whose sin / cos are used

    def use_trig(text):
        def sub(mo):
            arg = mo.group(2)
            if arg not in trig_args:
                trig_args.append(arg)
            return "%s%d_" % ("sn" if mo.group(1) == "sin" else "cn", trig_args.index(arg))
        return _re.sub(r"(?<![A-Za-z0-9_])(sin|cos)\((aux_\w+)\)", sub, text)

    def use_prod(text):  # the shared products by number: prod[k]
        return _re.sub(r"(?<![A-Za-z0-9_])%s(\d+)(?![0-9])" % cse.prefix, r"prod[\1]", text)

    def guard(v):  # NaN or Inf (not `v - v == 0`: where v is a product, FMA contraction turns v - v into its rounding error)
        return "if(!(fabs(%s) <= 1.7976931348623157e308)) bad[0]= 1;" % v

    outputs = []  # (member, index, rhs text, guarded)
    for i in range(len(self.tensor_tables["basis"])):
        outputs.append(("fxx", i, "basis[%d]" % i, False))
    # the box around the nominal input (limitsU): per input and side the tightest constraint, then minus u
    limit_code = {}
    for side, arr, cmp_, start in ((0, "lower", "<", "-INF"), (1, "upper", ">", "INF")):
        for j in range(self.m):
            code = "double lim_= %s; double bound_;\n" % start
            for c in self.cons:
                if c["input"] == j and (1 if c["sign"] > 0 else 0) == side:
                    code += "            bound_= %s; if(lim_%sbound_) lim_= bound_;\n" % (cexpr(self.csub(c["limit"], None)), cmp_)
            limit_code[(arr, j)] = code
            outputs.append((arr, j, "lim_ - u[%d]" % j, False))
    cse.use("first")
    groups = [self.jaco_items("fx", self.fx), self.jaco_items("fu", self.fu), self.grad_items("cx", self.Lx),
              self.hess_items("cxx", self.Lxx), self.grad_items("cu", self.Lu), self.hess_items("cuu", self.Luu),
              self.hess_items("cxu", self.Lxu)]
    for items in groups:
        for lhs, e in items:
            if self.is_time_var(e):
                m = _re.fullmatch(r"t->(\w+)\[(\d+)\]", lhs)
                outputs.append((m.group(1), int(m.group(2)), use_prod(use_trig(cexpr(self.csub(cse.rewrite(e), None)))), True))
    products = [(int(cse.names[prod].name[len(cse.prefix):]), use_trig(cexpr(self.csub(prod, None)))) for prod in cse.pools["first"]]
    basis = [use_trig(cexpr(self.csub(prod, None))) for prod in self.tensor_tables["basis"]]
    n_prod = max([k for k, _ in products] + [0]) + 1
    n_parts = (len(outputs) + OUT - 1) // OUT

    members = "".join("    double %s;\n" % q.name for q in tv_aux) or "    double unused_;\n"
    out = ("\n/* ---- additive: the time-varying entries of a step's derivative record in ILQG_DERIV_PARTS parts of\n"
           " * ILQG_DERIV_PART_OUT outputs each (batched back-ends that assemble records on chip and store whole cache lines; the\n"
           " * reference's solver never calls this).  Output o of the list ILQG_DERIV_OUTPUTS, X(member, index), is out[o %%\n"
           " * ILQG_DERIV_PART_OUT] of part o / ILQG_DERIV_PART_OUT.  ilqg_deriv_prepare: the step's auxiliaries, and — from sin and cos\n"
           " * of the auxiliaries, each evaluated once — the products the entries share (prod[]) and the products of\n"
           " * bp_tensor_basis (basis[]).  A NaN or Inf in a guarded value sets bad[0]. */\n"
           "#define ILQG_DERIV_PARTS %d\n#define ILQG_DERIV_PART_OUT %d\n#define ILQG_DERIV_NOUT %d\n#define ILQG_DERIV_NPROD %d\n"
           % (n_parts, OUT, len(outputs), n_prod))
    out += "#define ILQG_DERIV_OUTPUTS(X) " + " ".join("X(%s, %d)" % (mem, idx) for mem, idx, _, _ in outputs) + "\n"
    out += ("#ifndef ILQG_DERIV_SINCOS  /* a back-end may define it before including this file: SIN_ = sin(ARG_), COS_ = cos(ARG_) */\n"
            "#define ILQG_DERIV_SINCOS(ARG_, SIN_, COS_) ((SIN_)= sin(ARG_), (COS_)= cos(ARG_))\n#endif\n"
            "#ifndef ILQG_DERIV_PREPARE_FN  /* ... and the two functions' storage class / attributes */\n#define ILQG_DERIV_PREPARE_FN static\n#endif\n"
            "#ifndef ILQG_DERIV_PART_FN\n#define ILQG_DERIV_PART_FN static\n#endif\n"
            "#ifndef ILQG_DERIV_CASE  /* ... and what stands at the head of every part */\n#define ILQG_DERIV_CASE(q)\n#endif\n")
    out += "typedef struct {\n" + members + "} ilqg_deriv_aux_t;\n"
    out += ("ILQG_DERIV_PREPARE_FN void ilqg_deriv_prepare(ilqg_deriv_aux_t *t, double prod[], double basis[], int bad[], const double *x, const double *u, int k, double **p, int N) {\n")
    # products made of sin / cos values go to prod[]; the others (each used by a few entries of one array) are evaluated in
    # the part that uses them, so that they are not alive from here to there
    in_prepare = {kq for kq, text in products if _re.search(r"(sn|cn)\d+_", text)}
    slot = {kq: i for i, kq in enumerate(sorted(in_prepare))}  # prod[] holds these alone, densely
    # Every value is evaluated right in front of its first use — an auxiliary in front of the first sin / cos of it (its
    # assignment is the one of calcXVariableAux / calcXUVariableAux; an auxiliary defined through others pulls those in
    # first), a sin / cos pair in front of the first product of it — so that few of the 32 + 64 intermediate values are
    # alive at a time.
    aux_by_macro = {self.macro_name(q): q for q in tv_aux}
    done_aux, done_trig = set(), set()

    def need_aux(nm):
        text = ""
        if nm in done_aux or nm not in aux_by_macro:
            return text
        done_aux.add(nm)
        rhs = cexpr(self.csub(D.defs[aux_by_macro[nm]], None))
        for dep in _re.findall(r"(?<![A-Za-z0-9_])(aux_\w+)", rhs):
            text += need_aux(dep)
        return text + "    %s= %s;\n    %s\n" % (nm, rhs, guard(nm))

    def need_trig(text_using):
        text = ""
        for j in sorted({int(v) for v in _re.findall(r"(?:sn|cn)(\d+)_", text_using)}):
            if j not in done_trig:
                done_trig.add(j)
                text += need_aux(trig_args[j])
                text += "    double sn%d_, cn%d_;\n    ILQG_DERIV_SINCOS(%s, sn%d_, cn%d_);\n" % (j, j, trig_args[j], j, j)
        return text

    body = ""
    order = [("prod[%d]" % slot[kq], text) for kq, text in products if kq in in_prepare] + [("basis[%d]" % i, text) for i, text in enumerate(basis)]
    # (by the first sin / cos pair a value needs: the products of one pair of auxiliaries behind each other)
    order.sort(key=lambda it: min([int(v) for v in _re.findall(r"(?:sn|cn)(\d+)_", it[1])] or [0]))
    for lhs, text in order:
        body += need_trig(text) + "    %s= %s;\n    %s\n" % (lhs, text, guard(lhs))
    for nm in aux_by_macro:  # (auxiliaries nothing above needed: the struct is filled in any case)
        body += need_aux(nm)
    out += body + "}\n"
    out += ("ILQG_DERIV_PART_FN void ilqg_deriv_part(int part, double out[], int bad[], const ilqg_deriv_aux_t *t, const double prod[], const double basis[], const double *x, const double *u, int k, double **p, int N) {\n"
            "    switch(part) {\n")
    for q in range(n_parts):
        mine = outputs[q * OUT:(q + 1) * OUT]
        out += "    case %d: { ILQG_DERIV_CASE(%d)\n" % (q, q)
        local = []
        for _, _, rhs, _ in mine:
            for kq in _re.findall(r"prod\[(\d+)\]", rhs):
                if int(kq) not in in_prepare and int(kq) not in local:
                    local.append(int(kq))
        text_of = dict(products)
        for kq in sorted(local):
            out += "        const double pl%d_= %s;\n        %s\n" % (kq, text_of[kq], guard("pl%d_" % kq))
        mine = [(mem, idx, _re.sub(r"prod\[(\d+)\]", lambda mo: "prod[%d]" % slot[int(mo.group(1))] if int(mo.group(1)) in in_prepare else "pl%s_" % mo.group(1), rhs), g)
                for mem, idx, rhs, g in mine]
        for j, (mem, idx, rhs, guarded) in enumerate(mine):
            if (mem, idx) in limit_code:
                out += "        { %s            out[%d]= %s; }\n" % (limit_code[(mem, idx)], j, rhs)
            else:
                out += "        out[%d]= %s;\n" % (j, rhs)
                if guarded and not _re.fullmatch(r"-?[0-9.eE+-]+", rhs):
                    out += "        %s\n" % guard("out[%d]" % j)
        out += "        } break;\n"
    out += "    default: break;\n    }\n}\n"
    return out.replace("#define ILQG_DERIV_NPROD %d\n" % n_prod, "#define ILQG_DERIV_NPROD %d\n" % max(1, len(slot)))


Emitter.emit_deriv_parts = _emit_deriv_parts


def load_problem(path):
    spec = importlib.util.spec_from_file_location("ilqg_problem_def", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(Problem)


def main(argv):
    plain = "--plain" in argv
    argv = [a for a in argv if a != "--plain"]
    if len(argv) != 3:
        print(__doc__)
        return 2
    prob = load_problem(argv[1])
    em = Emitter(prob, plain=plain)
    os.makedirs(argv[2], exist_ok=True)
    with open(os.path.join(argv[2], "iLQG_problem.h"), "w") as f:
        f.write(em.problem_h())
    with open(os.path.join(argv[2], "iLQG_func.c"), "w") as f:
        f.write(em.func_c())
    print("wrote %s/{iLQG_problem.h,iLQG_func.c}: n=%d m=%d params=%s aux=%s" %
          (argv[2], em.n, em.m, em.param_names, [s.name for s in em.D.order]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
