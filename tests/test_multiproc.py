"""N > 1 path on CPU: two processes (gloo), each owns a contiguous shard of the trajectory batch,
no exchange except the single gather of costs — the same helpers bench.py uses with RCCL.
The per-shard work is done by the product when a GPU is present and by the CPU oracle otherwise (no GPU in the
build container); what is under test is the sharding and the collective."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_package

PER_RANK, ITERS = 3, 3


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from conftest import load_package as lp
    pkg = lp()
    from oracle.harness import CAR_PARAMS, Driver, lib_path
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    pkg.dist.init("gloo", rank, world)
    first = pkg.dist.shard_first(rank, PER_RANK)
    x0, u0 = pkg.synth.car_batch(PER_RANK, first=first)
    if pkg.ilqg.Problem("carparking", 0).device_count() > 0:
        # a GPU is present: the shard goes through the PRODUCT (both ranks share device 0 here)
        s = pkg.ilqg.BatchSolver("carparking", 0, batch=PER_RANK, n_hor=500, params=pkg.ilqg.CAR_PARAMS, opts=dict(max_iter=ITERS))
        s.init(x0, u0)
        s.iterate(ITERS)
        cost = s.scalar("cost")
        s.close()
    else:
        d = Driver(lib_path("oracle"), 500, CAR_PARAMS, dict(max_iter=ITERS))
        cost, _, _ = d.solve_many(x0, u0, 1)
    # the timed-region protocol of bench.py: barrier on both sides, the job's time is the slowest rank's, the value
    # is the work of ALL ranks over it
    pkg.dist.barrier(world)
    mine = 0.25 * (rank + 1)  # (a made-up duration per rank: rank 1 is the slower one)
    dt = pkg.dist.max_over_ranks(mine, world)
    assert dt == 0.25 * world
    assert pkg.dist.whole_job_rate(ITERS, dt, world) == ITERS * world / dt
    pkg.dist.barrier(world)
    allc = pkg.dist.gather_costs(torch.from_numpy(cost), rank, world)
    if rank == 0:
        np.save(out_path, allc.numpy())
    else:
        assert allc is None
    torch.distributed.destroy_process_group()


def test_two_rank_shards_and_single_gather(oracle_built, tmp_path):
    import torch.multiprocessing as mp
    from oracle.harness import CAR_PARAMS, Driver, lib_path
    pkg = load_package()
    out = str(tmp_path / "costs.npy")
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    x0, u0 = pkg.synth.car_batch(2 * PER_RANK)  # the whole batch in one process
    d = Driver(lib_path("oracle"), 500, CAR_PARAMS, dict(max_iter=ITERS))
    want, _, _ = d.solve_many(x0, u0, 1)
    assert got.shape == (2 * PER_RANK,)
    # rank r holds trajectories [r*PER_RANK, (r+1)*PER_RANK), in order (bit-equal on the CPU; where the shards ran on
    # a GPU they differ from the FMA-free checker by rounding)
    assert np.allclose(got, want, rtol=1e-9, atol=0) if pkg.ilqg.Problem("carparking", 0).device_count() > 0 else np.array_equal(got, want)


def test_generator_is_shard_invariant():
    synth = load_package().synth
    xa, ua = synth.car_batch(10)
    xb, ub = synth.car_batch(4, first=6)
    assert np.array_equal(xa[6:], xb) and np.array_equal(ua[6:], ub)
