"""Reproducible synthetic inputs for the batched CarParking workloads.

Counter-based generator: the splitmix64 finaliser applied to
seed + 0x9E3779B97F4A7C15 * (1 + b*2^20 + k*16 + i) gives a 53-bit uniform in
(0,1); pairs are turned into normals with Box-Muller.  Any element (b, k, i) can
be regenerated without files.  The arrays produced here are handed unchanged to
the CPU checker and to the GPU path, so both see bit-identical inputs.
"""
import numpy as np

SEED = 20261003
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(b, k, i, seed=SEED):
    """uniform(0,1) for integer arrays b (trajectory), k (time step), i (channel < 16)"""
    b, k, i = np.broadcast_arrays(np.asarray(b, np.uint64), np.asarray(k, np.uint64), np.asarray(i, np.uint64))
    with np.errstate(over="ignore"):
        ctr = np.uint64(1) + b * np.uint64(1 << 20) + k * np.uint64(16) + i
        z = _splitmix(np.uint64(seed) + _GOLD * ctr)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(b, k, i, seed=SEED):
    """standard normal from channels (i, i+8) of the same (b, k) cell"""
    u1 = uniform(b, k, i, seed)
    u2 = uniform(b, k, np.asarray(i) + 8, seed)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def car_batch(batch, n_hor=500, first=0, seed=SEED):
    """x0 [batch,4] and u0 [batch,n_hor,2] for trajectories first..first+batch-1.

    x0_b = (1+.5 xi1, 1+.5 xi2, 3pi/2+.5 xi3, .2 xi4), xi ~ U(-1,1);
    u0 = 0.1 N(0,1)  (reference examples/CarParking/testCar.m:16-17, randomised
    as SURVEY.md §8(d) config 2 prescribes).  Time step index n_hor is used for
    the x0 draws so they never collide with control draws.
    """
    b = np.arange(first, first + batch, dtype=np.uint64)
    xi = 2.0 * uniform(b[:, None], n_hor, np.arange(4)[None, :], seed) - 1.0
    x0 = np.array([1.0, 1.0, 1.5 * np.pi, 0.0]) + xi * np.array([0.5, 0.5, 0.5, 0.2])
    k = np.arange(n_hor, dtype=np.uint64)
    u0 = 0.1 * normal(b[:, None, None], k[None, :, None], np.arange(2)[None, None, :], seed)
    return np.ascontiguousarray(x0), np.ascontiguousarray(u0)


def car_single(n_hor=500, seed=SEED):
    """the reference demo's fixed x0 (testCar.m:16) with seeded controls"""
    _, u0 = car_batch(1, n_hor, first=0, seed=seed)
    return np.array([1.0, 1.0, 1.5 * np.pi, 0.0]), u0[0]


def synth16_batch(batch, n_hor, first=0, seed=SEED):
    """x0 [batch,16] ~ 0.5 U(-1,1), u0 [batch,n_hor,8] = 0.1 N(0,1) for the synthetic n=16, m=8 problem
    (SURVEY.md 8(d) config 5); same counter-based generator, separate streams from the car inputs"""
    b = np.arange(first, first + batch, dtype=np.uint64)
    x0 = 0.5 * (2.0 * uniform(b[:, None], n_hor + 1, np.arange(16)[None, :], seed + 1) - 1.0)
    k = np.arange(n_hor, dtype=np.uint64)
    u0 = 0.1 * normal(b[:, None, None], k[None, :, None], np.arange(8)[None, None, :], seed + 2)
    return np.ascontiguousarray(x0), np.ascontiguousarray(u0)


# parameters of the synthetic problem (problems/defs/synth16x8.py)
SYNTH16_PARAMS = dict(h=[0.05], c=[0.8], px=[0.1], ru=[0.05] * 8, qx=[0.02 + 0.01 * i for i in range(16)],
                      qf=[1.0 + 0.1 * i for i in range(16)], lim=[-1.0, 1.0])
