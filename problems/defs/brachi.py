"""Brachistochrone with a terminal equality constraint (reference examples/Brachistochrone/optDefBrachi.mac:1-13):
one state y (height, negative below the start), one input dy (slope), x advances by dx per step.  The running cost is
the travel time over one segment, the closed form of  integrate(sqrt((1+dy^2)/(2*g*abs(y+x_*dy))), x_, 0, dx)
for y < 0, dy < 0 (the reference's assumptions); hfe[1] = y - yf ties the end point down through the
augmented-Lagrangian multiplier path."""
import sympy as sp


def build(Problem):
    P = Problem("Brachi")
    (y,) = P.states("y")
    (dy,) = P.inputs("dy")
    dx = P.scalar("dx")
    g = P.scalar("g")
    yf = P.scalar("yf")
    P.f = [y + dy * dx]
    # int_0^dx (a + b x)^(-1/2) dx = 2 (sqrt(a + b dx) - sqrt(a)) / b with a = -y, b = -dy
    P.L = sp.sqrt((1 + dy**2) / (2 * g)) * 2 * (sp.sqrt(-y - dy * dx) - sp.sqrt(-y)) / (-dy)
    P.F = sp.Integer(0)
    P.hfe = [y - yf]
    P.fast = True
    return P
