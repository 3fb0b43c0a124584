"""Synthetic problem of BASELINE.json config 5 / SURVEY.md §8(d): n = 16 states, m = 8 inputs,
smooth nonlinear dynamics with DENSE first derivatives and non-zero second derivatives
fxx, fuu, fxu (so FULL_DDP = 1 moves 4 800 extra doubles per step), box limits on all inputs.

    f_i = x_i + h * ( sum_j A_ij x_j + sum_l B_il u_l + c * sin(s1_i) * cos(s2_i) )
    s1_i = sum_j W1_ij x_j        s2_i = sum_l W2_il u_l        (auxiliaries)
    L = sum_l ru_l u_l^2 + sum_i qx_i (sqrt(x_i^2 + px^2) - px)
    F = sum_i qf_i x_i^2

A, B, W1, W2 are fixed numeric matrices (seeded below) baked into the generated code."""
import numpy as np
import sympy as sp

N_X, N_U = 16, 8


def build(Problem):
    P = Problem("Synth16x8")
    P.fast = True  # skip sympy.simplify on the ~5 000 tensor entries
    P.cse = True   # name the products the entries share; factored tensor tables for the batched back-end
    x = P.states(" ".join("x%d" % i for i in range(N_X)))
    u = P.inputs(" ".join("u%d" % i for i in range(N_U)))
    h = P.scalar("h")
    c = P.scalar("c")
    px = P.scalar("px")
    ru = P.vector("ru", N_U)
    qx = P.vector("qx", N_X)
    qf = P.vector("qf", N_X)
    lim = P.vector("lim", 2)

    rng = np.random.default_rng(16082026)
    r3 = lambda a: sp.Float(round(float(a), 3))
    A = -1.2 * np.eye(N_X) + 0.15 * rng.standard_normal((N_X, N_X))  # stable linear part (N = 1000 horizons)
    Bm = 0.5 * rng.standard_normal((N_X, N_U))
    W1 = 0.4 * rng.standard_normal((N_X, N_X))
    W2 = 0.6 * rng.standard_normal((N_X, N_U))

    s1 = [P.auxiliary("s1_%d" % i, sum(r3(W1[i, j]) * x[j] for j in range(N_X))) for i in range(N_X)]
    s2 = [P.auxiliary("s2_%d" % i, sum(r3(W2[i, l]) * u[l] for l in range(N_U))) for i in range(N_X)]
    P.f = [x[i] + h * (sum(r3(A[i, j]) * x[j] for j in range(N_X)) + sum(r3(Bm[i, l]) * u[l] for l in range(N_U))
                       + c * sp.sin(s1[i]) * sp.cos(s2[i])) for i in range(N_X)]
    P.L = sum(ru[l] * u[l]**2 for l in range(N_U)) + sum(qx[i] * (sp.sqrt(x[i]**2 + px**2) - px) for i in range(N_X))
    P.F = sum(qf[i] * x[i]**2 for i in range(N_X))
    P.h = []
    for l in range(N_U):
        P.h.append(-u[l] + lim[0])
        P.h.append(u[l] - lim[1])
    return P
