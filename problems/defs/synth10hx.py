"""n = 10, m = 3 with FACTORED tensors AND input limits that depend on the state: the combination no other
shipped problem has — wave mapping (n > 8), FULL_DDP = 1 from the factored tensor tables, and the constraint-gradient terms
of the feedback gains (back_pass.c:186-199; do_limits / do_hx, genenerator_main.mac:419-447).  k_derivs_wave's factored
instantiation must then run the callbacks on the record itself (limitsU() stores the limits' signs and gradients, which the
row-mapped backward step reads from it), not on a private element.

    f_i = x_i + h * ( sum_j A_ij x_j + sum_l B_il u_l + c * sin(s1_i) * cos(s2_i) ),   s1 = W1 x, s2 = W2 u
    L, F as synth16x8;   u0 <= lim + x1 / 2,   -u0 <= lim,   |u1| <= lim,   u2 <= lim,   -u2 <= lim + x0^2 / 5"""
import numpy as np
import sympy as sp

N_X, N_U = 10, 3


def build(Problem):
    P = Problem("Synth10Hx")
    P.fast = True
    P.cse = True
    x = P.states(" ".join("x%d" % i for i in range(N_X)))
    u = P.inputs(" ".join("u%d" % i for i in range(N_U)))
    h = P.scalar("h")
    c = P.scalar("c")
    px = P.scalar("px")
    ru = P.vector("ru", N_U)
    qx = P.vector("qx", N_X)
    qf = P.vector("qf", N_X)
    lim = P.scalar("lim")

    rng = np.random.default_rng(10032026)
    r3 = lambda a: sp.Float(round(float(a), 3))
    A = -1.2 * np.eye(N_X) + 0.2 * rng.standard_normal((N_X, N_X))
    Bm = 0.8 * rng.standard_normal((N_X, N_U))
    W1 = 0.5 * rng.standard_normal((N_X, N_X))
    W2 = 0.7 * rng.standard_normal((N_X, N_U))
    s1 = [P.auxiliary("s1_%d" % i, sum(r3(W1[i, j]) * x[j] for j in range(N_X))) for i in range(N_X)]
    s2 = [P.auxiliary("s2_%d" % i, sum(r3(W2[i, l]) * u[l] for l in range(N_U))) for i in range(N_X)]
    P.f = [x[i] + h * (sum(r3(A[i, j]) * x[j] for j in range(N_X)) + sum(r3(Bm[i, l]) * u[l] for l in range(N_U))
                       + c * sp.sin(s1[i]) * sp.cos(s2[i])) for i in range(N_X)]
    P.L = sum(ru[l] * u[l]**2 for l in range(N_U)) + sum(qx[i] * (sp.sqrt(x[i]**2 + px**2) - px) for i in range(N_X))
    P.F = sum(qf[i] * x[i]**2 for i in range(N_X))
    P.h = [u[0] - (lim + x[1] / 2),
           -u[0] - lim,
           u[1] - lim,
           -u[1] - lim,
           u[2] - lim,
           -u[2] - (lim + x[0]**2 / 5)]
    return P
