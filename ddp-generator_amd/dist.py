"""Multi-GPU plumbing: trajectories are independent, so the batch shards across ranks with no
data-path collective; the only exchange is ONE gather of the per-trajectory costs
(BASELINE.json north_star: "a single RCCL gather of costs over xGMI").  One process per GPU,
`torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
import os


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) when absent"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_first(rank, per_rank):
    """global index of the first trajectory owned by `rank` (contiguous blocks)"""
    return rank * per_rank


def init(backend, rank, world, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def barrier(world, device_sync=None):
    """both sides of a timed region: every rank has arrived AND its device has drained (bench.py's contract)"""
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if device_sync is not None:
        device_sync()


def max_over_ranks(seconds, world, device=None):
    """the job's time for a region = the slowest rank's (a whole-job rate divides the work of ALL ranks by it)"""
    if world == 1:
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank, seconds_max, world):
    """`value` of the bench line: the units ALL ranks processed / the slowest rank's time (weak scaling)"""
    return units_per_rank * world / seconds_max


def gather_costs(cost, rank, world, dst=0):
    """the path's single collective: every rank's cost vector [per_rank] -> rank `dst` [world*per_rank]
    (None on the other ranks).  `cost` may be a zero-copy view of the solver's device memory."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return cost
    out = torch.empty(cost.numel() * world, dtype=cost.dtype, device=cost.device) if rank == dst else None
    chunks = list(out.chunk(world)) if rank == dst else None
    dist.gather(cost, chunks, dst=dst)
    return out


def device_view(ptr, n, device, fallback=None):
    """zero-copy torch view of `n` doubles of device memory owned by the solver; if the array-interface
    import is unavailable in this torch build, `fallback()` must return the values as a numpy array"""
    import torch

    class _Arr:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

    try:
        return torch.as_tensor(_Arr(), device=device)
    except Exception:
        if fallback is None:
            raise
        return torch.from_numpy(fallback()).to(device)
