"""Second synthetic n = 16, m = 8 problem: the dynamics of synth16x8 plus PAIRWISE STATE PRODUCTS
inside the nonlinearity, so that the second derivatives of f_i are no longer a number times ONE product shared by the
slice — the structure tools/gen_problem.py's _factor_tensors() finds in synth16x8 and the factored tensor tables rest on:

    f_i = x_i + h * ( sum_j A_ij x_j + sum_l B_il u_l + c * sin(s1_i) * cos(s2_i)
                      + e * x_{p(i)} * x_{q(i)} * cos(u_{r(i)}) )
    s1_i = sum_j W1_ij x_j        s2_i = sum_l W2_il u_l        (auxiliaries)

d2 f_i / dx_p dx_q gets the extra term e cos(u_r), d2 f_i / dx_p du_r the term -e x_q sin(u_r), d2 f_i / du_r du_r the
term -e x_p x_q cos(u_r): three more products per slice, each in a few entries only.  No tables can be made; the record
carries the tensors fxx / fuu / fxu (47.9 KB per step) and the backward step contracts them from HBM
(back_pass.c:95-131) — the path every Maxima-generated n = 16 pair takes.  Same costs, limits and parameters as
synth16x8 (plus e)."""
import numpy as np
import sympy as sp

N_X, N_U = 16, 8


def build(Problem):
    P = Problem("Synth16Pair")
    P.fast = True
    P.cse = True   # (shared products are still named in bp_derivsL; the factoring fails and no tables are emitted)
    x = P.states(" ".join("x%d" % i for i in range(N_X)))
    u = P.inputs(" ".join("u%d" % i for i in range(N_U)))
    h = P.scalar("h")
    c = P.scalar("c")
    e = P.scalar("e")
    px = P.scalar("px")
    ru = P.vector("ru", N_U)
    qx = P.vector("qx", N_X)
    qf = P.vector("qf", N_X)
    lim = P.vector("lim", 2)

    rng = np.random.default_rng(16082026)  # (the matrices of synth16x8)
    r3 = lambda a: sp.Float(round(float(a), 3))
    A = -1.2 * np.eye(N_X) + 0.15 * rng.standard_normal((N_X, N_X))
    Bm = 0.5 * rng.standard_normal((N_X, N_U))
    W1 = 0.4 * rng.standard_normal((N_X, N_X))
    W2 = 0.6 * rng.standard_normal((N_X, N_U))

    s1 = [P.auxiliary("s1_%d" % i, sum(r3(W1[i, j]) * x[j] for j in range(N_X))) for i in range(N_X)]
    s2 = [P.auxiliary("s2_%d" % i, sum(r3(W2[i, l]) * u[l] for l in range(N_U))) for i in range(N_X)]
    pair = lambda i: x[(i + 3) % N_X] * x[(i + 7) % N_X] * sp.cos(u[i % N_U])
    P.f = [x[i] + h * (sum(r3(A[i, j]) * x[j] for j in range(N_X)) + sum(r3(Bm[i, l]) * u[l] for l in range(N_U))
                       + c * sp.sin(s1[i]) * sp.cos(s2[i]) + e * pair(i)) for i in range(N_X)]
    P.L = sum(ru[l] * u[l]**2 for l in range(N_U)) + sum(qx[i] * (sp.sqrt(x[i]**2 + px**2) - px) for i in range(N_X))
    P.F = sum(qf[i] * x[i]**2 for i in range(N_X))
    P.h = []
    for l in range(N_U):
        P.h.append(-u[l] + lim[0])
        P.h.append(u[l] - lim[1])
    return P
