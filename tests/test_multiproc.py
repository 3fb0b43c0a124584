"""N > 1 path on CPU: two processes (gloo), each owns a contiguous shard of the trajectory batch,
no exchange except the single gather of costs — the same helpers bench.py uses with RCCL.
The per-shard work is done by the product when a GPU is present and by the CPU oracle otherwise (no GPU in the
build container); what is under test is the sharding and the collective."""
import json
import os
import subprocess
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


def _run_bench(argv, nproc=0, port=0):
    """bench.py as the driver launches it (nproc > 1: torch.distributed.run, one process per rank); its JSON line"""
    import json
    import subprocess
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py")] + argv
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE line, from rank 0
    assert r.stdout.strip() == lines[0], r.stdout[-2000:]  # and nothing else on stdout (console lines of the C checkers)
    assert len(lines[0]) < 6000, len(lines[0])  # what the driver parses; the full report is bench_detail.json
    out = json.loads(lines[0])
    assert out["detail"] == "bench_detail.json" and os.path.exists(os.path.join(ROOT, out["detail"]))
    return out


def test_bench_main_with_two_ranks_protocol_rehearsal():
    """bench.py's own N > 1 branch (shard offsets, barriers, the single gather, max over ranks, rank-0 JSON) with
    world = 2 on CPU ranks: --rehearse-protocol puts a stand-in without numerics where the solver stands"""
    port = 29900 + (os.getpid() % 90)
    out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "96", "--rehearse-protocol"], nproc=2, port=port)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["value"] is None and "rehearsal" in out  # not a measurement, and says so
    assert out["gathered_costs_in_order"] is True       # rank r's shard sits at [r * 96, (r + 1) * 96) on rank 0
    assert out["collective"] == {"backend": "gloo", "tensors": "host memory", "doubles_per_rank": 96, "gathered_on_rank_0": 192}
    assert out["config"]["batch_per_gpu"] == 96 and "x 2 GPU" in out["config"]["workload"]
    assert out["roofline"]["bound"] == "hbm" and out["ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_main_with_two_ranks_on_one_gpu():
    """the same branch with the PRODUCT under it: two ranks (gloo, collective through host memory) share device 0, each
    advances its shard; the line is a real (small) measurement and the gathered costs are those of one 2 x 512 batch"""
    port = 29800 + (os.getpid() % 90)
    out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512", "--backend", "gloo", "--all-on-device", "0"],
                     nproc=2, port=port)
    assert out["n_gpus"] == 2 and out["value"] > 0 and "rehearsal" not in out
    # the line of an N > 1 run is complete: the CPU baseline (rank 0's host cores) and a per-kernel roofline of rank 0's
    # own launches (its shard), not the iteration-level stand-in
    assert out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] in ("reference", "port")
    rf = out["roofline"]
    assert rf["kernel"] == "k_backward[fused derivs]" and rf["bound"] == "hbm" and 0 < rf["frac"] < 1.5
    assert rf["launches"] >= 3 and rf["trajectories_per_launch"] * out["config"]["stream_groups"] == 512
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-9 * rf["achieved"]
    import json
    detail = json.load(open(os.path.join(ROOT, out["detail"])))  # the full report carries the line's figures and more
    assert detail["value"] == out["value"] and detail["roofline"]["dominant_launch"]["launches"] == rf["launches"]
    assert abs(out["value"] - 2 * out["per_gpu_iterations_per_s"]) < 1e-9 * out["value"]  # whole job = 2 shards
    assert out["collective"]["gathered_on_rank_0"] == 1024 and out["trajectories_still_active"] == 512
    pkg = load_package()
    x0, u0 = pkg.synth.car_batch(1024)
    s = pkg.ilqg.BatchSolver("carparking", 0, batch=1024, n_hor=500, params=pkg.ilqg.CAR_PARAMS, opts=dict(max_iter=4))
    s.init(x0, u0)
    s.iterate(3)
    want = float(s.scalar("cost").mean())
    s.close()
    assert abs(out["cost_mean_after_window"] - want) <= 1e-12 * abs(want)


@pytest.mark.gpu
def test_bench_guards_the_device_ordinal():
    """LOCAL_RANK is a device ordinal only while the rank sees every GPU: with ONE visible device a rank's ordinal is 0
    (a launcher that narrows HIP_VISIBLE_DEVICES per rank), and an ordinal beyond the visible devices otherwise is an
    error that says so instead of a HIP failure somewhere inside"""
    import torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ, LOCAL_RANK=str(ndev + 2), RANK="0", WORLD_SIZE="1")
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--batch", "256", "--no-unfused", "--no-cpu-baseline"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    if ndev == 1:
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert out["config"]["device_ordinal_of_rank_0"] == 0 and "one visible device" in out["config"]["device_note"]
    else:
        assert r.returncode != 0 and "device(s) are visible" in r.stderr
    r = subprocess.run(args + ["--all-on-device", str(ndev + 2)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "device(s) are visible" in r.stderr and "Traceback" not in r.stderr


@pytest.mark.gpu
def test_bench_single_process_eight_loopback_shards():
    """bench.py --single-process --gpus 8 over the device list [0] * 8 (ilqg_multi_*: eight shards, offsets, per-shard
    contexts, the cost hand-over) emits the line"""
    out = _run_bench(["--single-process", "--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--batch", "256", "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == 8 and out["value"] > 0 and out["config"]["devices"] == [0] * 8
    assert abs(out["value"] - 8 * out["per_gpu_iterations_per_s"]) < 1e-9 * out["value"]
    assert out["trajectories_still_active"] == 8 * 256
