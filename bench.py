#!/usr/bin/env python3
"""Benchmark: batched iLQG iterations/s on CarParking (n=4, m=2, N=500), 65 536
trajectories per GPU, fp64 (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one lock-step iLQG iteration of the whole batch: calc_derivs of the
nominal trajectories, back_pass (with its lambda retries), line_search over all 8
step sizes, accept/reject bookkeeping — for every trajectory.  The timed window
is the FIRST K iterations after the initial roll-out (no CarParking trajectory
converges before iteration 50, so all trajectories are active throughout).
Warm-up iterations run on the same inputs and the solver is then re-initialised
(untimed), so the timed work is always iterations 1..K.

One process per GPU; trajectories are independent, so ranks share nothing but a
single RCCL gather of the per-trajectory costs at the end of the timed window
(weak scaling: 65 536 trajectories per GPU).  PyTorch is used only for process
rendezvous, the RCCL collective and device synchronisation.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel from its
HIP-event time measured inside this run; `cpu_baseline` times the CPU checker
(the reference's own sources when oracle/_ref was shipped, else the C port) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_HOR = 500
NX, NU = 4, 2
SXX, SUU, NXU = 10, 3, 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

# ALGORITHMIC bytes per time step and trajectory (doubles x 8 B)
ALG_BYTES = {
    "k_derivs": (NX + NU + (NX + NU + SXX + SUU + NXU + NX * NX + NXU + 2 * NU)) * 8,      # 61 dbl = 488 B
    "k_backward": ((NX + SXX + NU + SUU + NXU + NX * NX + NXU + 2 * NU + NU) + NU + NXU) * 8,  # 67 dbl = 536 B
    "k_rollout[search]": (NX + 2 * NU + NXU) * 8,                                          # 16 dbl read, shared by all alpha
    "k_rollout[winner]": (NX + NU) * 8,                                                     # 6 dbl written (winner only)
}
# derivatives evaluated inside the backward kernel: priced against the UNFUSED figure of the two
# kernels it replaces; what it actually moves is 6 dbl read + 10 written = 128 B
ALG_BYTES["k_backward[fused derivs]"] = ALG_BYTES["k_derivs"] + ALG_BYTES["k_backward"]
FUSED_MOVED_BYTES = (NX + NU) * 8 + (NX + 2 * NU + NXU) * 8
ITERATION_BYTES = 1200  # per step and trajectory, SURVEY.md §8(d)


def cpu_baseline(batch_per_gpu, iters, problem, fd, params, n_hor, make_inputs, budget_s=6.0):
    """CPU checker on a bounded sample of the same workload, one pthread per host core, each solving
    its share of the sample exactly as independent runs of the reference would (oracle/driver.c,
    drv_solve_many).  Returns the JSON object for `cpu_baseline`."""
    from oracle.harness import Driver, lib_path
    ref = lib_path("ref", problem, fd)
    kind = "reference" if os.path.exists(ref) else "port"
    path = ref if kind == "reference" else lib_path("oracle", problem, fd)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    d = Driver(path, n_hor, params, dict(max_iter=iters))
    # calibrate single-core speed on a few trajectories, then size the sample to the time budget
    ncal = 8 if problem == "carparking" else 1
    x0, u0 = make_inputs(ncal, n_hor)
    t0 = time.perf_counter()
    d.solve_many(x0, u0, 1)
    per_traj = max((time.perf_counter() - t0) / ncal, 1e-5)
    # threads on a loaded many-core host run ~2-3x slower than the single calibration thread: the budget is
    # sized for ~10-20 s of wall time
    sample = int(max(cores, min(32768, budget_s / per_traj * cores)))
    x0, u0 = make_inputs(sample, n_hor)
    t0 = time.perf_counter()
    cost, its, rc = d.solve_many(x0, u0, cores)
    dt = time.perf_counter() - t0
    d.close()
    traj_iters_per_s = float(its.sum()) / dt
    return {
        "value": traj_iters_per_s / batch_per_gpu,  # batched iterations/s of a whole-batch equivalent
        "unit": "iterations/s (%d-trajectory batch equivalent)" % batch_per_gpu,
        "cores": cores,
        "kind": kind,
        "single_core_ms_per_trajectory_iteration": 1e3 * per_traj / iters,
        "sample": "%d trajectories x %d iterations (same generator, trajectories 0..%d) in %.1f s on %d threads; "
                  "%.0f trajectory-iterations/s" % (sample, iters, sample - 1, dt, cores, traj_iters_per_s),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("car", "synth"), default="car",
                    help="car: BASELINE metric (CarParking n=4,m=2,N=500, 65 536 per GPU); synth: BASELINE config 5 "
                         "(n=16,m=8,N=1000, FULL_DDP=1, 16 384 per GPU, one wavefront per trajectory)")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU")
    ap.add_argument("--n-hor", type=int, default=None)
    ap.add_argument("--full-ddp", type=int, default=None)
    ap.add_argument("--mapping", choices=("auto", "wave"), default="auto",
                    help="wave: CarParking in the one-wavefront-per-trajectory build (comparison)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resweep", type=int, default=-1, help="-1: library default (off without multipliers)")
    ap.add_argument("--fuse-derivs", type=int, default=1)
    ap.add_argument("--ls-split", type=int, default=3)
    ap.add_argument("--no-unfused", action="store_true", help="skip the secondary run with materialised derivative records")
    ap.add_argument("--groups", type=int, default=0,
                    help="independent sets of trajectories advanced on separate HIP streams (0: library default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    from ddp_generator_amd import ilqg, synth

    rank, local, world = pkg.dist.env_world()
    if world > 1:
        pkg.dist.init("nccl", rank, world, torch.device("cuda", local))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    global N_HOR, NX, NU, SXX, SUU, NXU, ALG_BYTES, FUSED_MOVED_BYTES, ITERATION_BYTES
    car = args.workload == "car"
    problem = "carparking" if car else "synth16x8"
    fd = args.full_ddp if args.full_ddp is not None else (0 if car else 1)
    B = args.batch if args.batch is not None else (65536 if car else 16384)
    N_HOR = args.n_hor if args.n_hor is not None else (500 if car else 1000)
    K, W = args.steps, args.warmup
    if not car:
        NX, NU = 16, 8
        SXX, SUU, NXU = NX * (NX + 1) // 2, NU * (NU + 1) // 2, NX * NU
    rec = NX + SXX + NU + SUU + NXU + NX * NX + NXU + 2 * NU + (NX * (SXX + SUU + NXU) if fd else 0)
    ALG_BYTES = {  # SURVEY.md 8(d) formulas, doubles x 8 B per step and trajectory
        "k_derivs": (NX + NU + rec) * 8,
        "k_backward": (rec + NU + NU + NXU) * 8,
        "k_rollout[search]": (NX + 2 * NU + NXU) * 8,
        "k_rollout[winner]": (NX + NU) * 8,
    }
    ALG_BYTES["k_backward[fused derivs]"] = ALG_BYTES["k_derivs"] + ALG_BYTES["k_backward"]
    FUSED_MOVED_BYTES = (NX + NU) * 8 + (NX + 2 * NU + NXU) * 8  # reads (x_k,u_k), writes the packed record of step k
    ITERATION_BYTES = sum(ALG_BYTES[k] for k in ("k_derivs", "k_backward", "k_rollout[search]", "k_rollout[winner]"))
    first = pkg.dist.shard_first(rank, B)
    x0, u0 = synth.car_batch(B, N_HOR, first=first) if car else synth.synth16_batch(B, N_HOR, first=first)
    params = ilqg.CAR_PARAMS if car else synth.SYNTH16_PARAMS
    s = ilqg.BatchSolver(problem, fd, batch=B, n_hor=N_HOR, device=local, params=params,
                         opts=dict(max_iter=max(K, W) + 1, fuse_derivs=args.fuse_derivs, ls_split=args.ls_split),
                         strict=("wave" if args.mapping == "wave" else False), groups=args.groups)
    args.full_ddp = fd
    if args.resweep >= 0:
        s.set_option("resweep", args.resweep)
    s.init(x0, u0)
    if W > 0:
        s.iterate(W)
        s.sync()
        s.init(x0, u0)  # back to iteration 0: the timed window is always iterations 1..K
    # device buffer the solver's per-trajectory costs are copied into (device to device) for the collective
    cost_dev = torch.empty(B, dtype=torch.float64, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    s.timing(True)
    barrier()
    t0 = time.perf_counter()
    s.iterate(K)
    s.scalar_to_device("cost", cost_dev.data_ptr())  # synchronises the solver's streams
    # the single collective of the path: per-trajectory costs to rank 0 over RCCL/xGMI
    gathered = pkg.dist.gather_costs(cost_dev, rank, world)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    times = s.kernel_times()
    active = s.active()
    stream_groups = s.groups()
    cost = gathered.cpu().numpy() if world > 1 and rank == 0 else s.scalar("cost")

    # secondary, untimed for `value`: the same iterations with the derivative records materialised in
    # HBM (k_derivs + k_backward<0>), the two kernels the HBM roofline of SURVEY 8(d) was written for
    # It runs as ONE group of trajectories, so that a launch covers the whole batch and nothing else is on the GPU
    # while it is timed (the timed run above overlaps the kernels of several groups).
    unfused = {}
    if rank == 0 and args.fuse_derivs and not args.no_unfused and not s.problem.wave_mapping:
        s.close()
        s = ilqg.BatchSolver(problem, fd, batch=B, n_hor=N_HOR, device=local, params=params,
                             opts=dict(max_iter=max(K, W) + 1, fuse_derivs=0, ls_split=args.ls_split), groups=1)
        s.init(x0, u0)
        s.timing(True)
        s.iterate(5)
        s.sync()
        for kname, (n, ms) in s.kernel_times().items():
            if n and kname in ("k_derivs", "k_backward"):
                b_alg = ALG_BYTES[kname] * N_HOR * B
                unfused[kname] = {"avg_launch_ms": ms / n, "algorithmic_bytes_per_launch": b_alg,
                                  "achieved_GBs": b_alg / (ms / n * 1e-3) / 1e9,
                                  "frac_of_peak": b_alg / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "stream_groups": 1}
    if rank == 0:
        per_iter = {k: v[1] / max(1, K) for k, v in times.items() if v[0]}
        # the kernel the HBM roofline is about: the one that accounts for most ALGORITHMIC bytes of an iteration
        # (the backward pass incl. derivatives: 1 024 of 1 200 B per step and trajectory; the roll-outs read 128 B
        # shared by all step sizes and are fp64-VALU bound — their times are in kernels_ms_per_iteration)
        dominant = max((k for k in per_iter if k in ALG_BYTES), key=lambda k: ALG_BYTES[k] * times[k][0])
        n_launch, total_ms = times[dominant]
        avg_ms = total_ms / n_launch
        # per launch; in the wave mapping a kernel is launched once per chunk of trajectories per iteration
        alg_bytes = ALG_BYTES[dominant] * N_HOR * B * K / n_launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from rocprofv3 PMC passes (tools/collect_traffic.sh), if committed
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and B == 65536 and car and not s.problem.wave_mapping:
            traffic = json.load(open(tpath)).get(dominant, {}).get("hbm_bytes_per_launch")
        iter_bytes = ITERATION_BYTES * N_HOR * B
        out = {
            "metric": ("iLQG iterations/sec, 65k-batch CarParking (n=4,m=2,N=500)" if car else
                       "iLQG iterations/sec, batch %d synthetic problem (n=16,m=8,N=%d, FULL_DDP=%d)" % (B, N_HOR, fd)),
            "value": K / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * dt / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s batch=%d per GPU x %d GPU, 8-alpha line search, FULL_DDP=%d, "
                                   "first %d iterations after the initial roll-out" % ("CarParking" if car else "Synth16x8", B, world, args.full_ddp, K),
                       "batch_per_gpu": B, "n_hor": N_HOR, "n_x": NX, "n_u": NU, "full_ddp": args.full_ddp,
                       "mapping": ("one wavefront per trajectory" if s.problem.wave_mapping else
                                   "one lane per trajectory (64 trajectories per wavefront)"),
                       "fuse_derivs": args.fuse_derivs, "ls_split": args.ls_split, "resweep": args.resweep,
                       "stream_groups": stream_groups,
                       "parallelism": "batch sharded over %d GPU, one RCCL gather of costs" % world},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms, "launches": n_launch,
                         "note": "achieved = ALGORITHMIC bytes of SURVEY 8(d) per launch / average HIP-event "
                                 "time of a launch. For k_backward[fused derivs] that is the figure of the two "
                                 "kernels it replaces (k_derivs 488 B + k_backward 536 B per step and trajectory), "
                                 "as SURVEY 8(d) prescribes; the fused kernel itself moves 176 B per step and "
                                 "trajectory (48 B read, the 128 B record of the step written = traffic) and is bound by fp64 VALU issue, not by HBM. The batch "
                                 "advances as stream_groups sets of trajectories on separate streams: a launch "
                                 "covers one set and shares the GPU with the kernels of the others while it is "
                                 "timed. See unfused_kernels for the HBM-bound kernels measured alone, and "
                                 "iteration_roofline for the whole iteration.",
                         "moved_bytes_per_launch": (FUSED_MOVED_BYTES * N_HOR * B * K / n_launch) if "fused" in dominant else alg_bytes},
            "unfused_kernels": unfused,
            "iteration_roofline": {"algorithmic_bytes_per_iteration": iter_bytes,
                                   "achieved_GBs": iter_bytes * (K / dt) / 1e9 / world * 1.0,
                                   "frac_of_peak": iter_bytes * (K / dt) / world / 1e9 / HBM_PEAK_GBS},
            "kernels_ms_per_iteration": per_iter,
            "trajectories_still_active": int(active),
            "cost_mean_after_window": float(cost.mean()),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, K, problem, fd, params, N_HOR,
                                               synth.car_batch if car else synth.synth16_batch)
        print(json.dumps(out))
    s.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
