"""Car-parking problem of Tassa's iLQG demo, as defined by the reference in
examples/CarParking/optDefCar.mac:1-19 (states x_, y_, t, v; inputs w, a;
auxiliary rolling distance s; smooth-abs costs; box limits on both inputs).
Parameter values used by the demo are in examples/CarParking/testCar.m:2-11."""
import sympy as sp


def build(Problem):
    P = Problem("CarParking")
    x_, y_, t, v = P.states("x_ y_ t v")
    w, a = P.inputs("w a")
    d = P.scalar("d")
    h = P.scalar("h")
    cf = P.vector("cf", 4)
    cu = P.vector("cu", 2)
    cx = P.vector("cx", 2)
    pf = P.vector("pf", 4)
    px = P.vector("px", 2)
    limW = P.vector("limW", 2)
    limA = P.vector("limA", 2)

    # distance rolled by the rear axle in one step (optDefCar.mac:4)
    s = P.auxiliary("s", d + h * v * sp.cos(w) - sp.sqrt(d**2 - (h * v * sp.sin(w))**2))
    P.f = [x_ + s * sp.cos(t),
           y_ + s * sp.sin(t),
           t + sp.asin(sp.sin(w) * h * v / d),
           v + h * a]

    def sabs(z, e):  # smooth |z| (optDefCar.mac:11)
        return sp.sqrt(z**2 + e**2) - e

    P.F = (cf[0] * sabs(x_, pf[0]) + cf[1] * sabs(y_, pf[1]) + cf[2] * sabs(t, pf[2])
           + cf[3] * sabs(v, pf[3]) + cx[0] * sabs(x_, px[0]) + cx[1] * sabs(y_, px[1]))
    P.L = cu[0] * w**2 + cu[1] * a**2 + cx[0] * sabs(x_, px[0]) + cx[1] * sabs(y_, px[1])
    P.h = [-w + limW[0], w - limW[1], -a + limA[0], a - limA[1]]
    return P
