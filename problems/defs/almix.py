"""Test problem with all four kinds of augmented-Lagrangian constraints at once (the reference's demos use
hfe and hli only): running equality hle and inequality hli, final equality hfe and inequality hfi
(genenerator_main.mac:46-124), two constraints of one kind so that the multiplier structs hold arrays, an
auxiliary feeding a constraint, and a box-limited input next to them.

3 states (position p, velocity v, towed position q), 2 inputs (force a, coupling s); only + - * / and sqrt,
so the -ffp-contract=off device build reproduces the CPU bit for bit."""
import sympy as sp


def build(Problem):
    P = Problem("AlMix")
    p, v, q = P.states("p v q")
    a, s = P.inputs("a s")
    h = P.scalar("h")
    cu = P.vector("cu", 2)
    cx = P.vector("cx", 3)
    cf = P.vector("cf", 3)
    lim = P.vector("lim", 2)
    tgt = P.vector("tgt", 3)   # final position, final position bound, velocity bound
    vref = P.per_step("vref")  # per-step coupling target

    gap = P.auxiliary("gap", p - q)
    P.f = [p + h * v,
           v + h * (a - gap / 4 - v**3 / 2),   # cubic drag: non-zero second derivatives for FULL_DDP
           q + h * (s + gap / 2)]
    P.L = cu[0] * a**2 + cu[1] * s**2 + cx[0] * (sp.sqrt(gap**2 + 1) - 1) + cx[1] * v**2 + cx[2] * q**2
    P.F = cf[0] * (p - tgt[0])**2 + cf[1] * v**2 + cf[2] * (q - tgt[0])**2
    P.h = [a - lim[1], -a + lim[0]]
    P.hle = [s - vref * v / 2]                  # the coupling input follows the velocity
    P.hli = [v - tgt[2], -v - tgt[2]]           # |v| <= bound
    P.hfe = [v, gap]                            # at rest and closed up at the end
    P.hfi = [p - tgt[1]]                        # final position below a bound
    P.fast = True
    return P
