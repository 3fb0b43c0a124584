"""Brachistochrone with a running inequality constraint and a terminal equality on a per-time-step parameter
(reference examples/Brachistochrone/optDefBrachi_hli.mac:1-14): hli[1] = ymin[k] - y <= 0, hfe[1] = y - ymin[k]."""
import sympy as sp


def build(Problem):
    P = Problem("BrachiHli")
    (y,) = P.states("y")
    (dy,) = P.inputs("dy")
    dx = P.scalar("dx")
    g = P.scalar("g")
    ymin = P.per_step("ymin")
    P.f = [y + dy * dx]
    P.L = sp.sqrt((1 + dy**2) / (2 * g)) * 2 * (sp.sqrt(-y - dy * dx) - sp.sqrt(-y)) / (-dy)
    P.F = sp.Integer(0)
    P.hli = [ymin - y]
    P.hfe = [y - ymin]
    P.fast = True
    return P
