"""One-process-per-GPU replica plumbing for the multi-GPU runs (torch.distributed only).

The hot path does not shard inside a sequence (the solve is latency-bound, a 512^3 volume fits
one GPU): independent sequences are assigned to ranks and run with NO data-path collective
(SURVEY.md §8e "replicas only").  torch.distributed (backend "nccl" == RCCL on the GPUs, "gloo"
in the CPU tests) is used for exactly three things: rendezvous, the barriers that bracket the
timed region, and the MAX-over-ranks of the elapsed time.
"""
import os
import time


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """Initialises the default process group when WORLD_SIZE > 1. Returns (rank, local, world)."""
    import torch.distributed as dist
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        kw = {}
        if backend is None:
            backend = "nccl" if device is not None and device.type == "cuda" else "gloo"
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return rank, local, world


def assign_sequences(n_sequences, world, rank):
    """Sequence ids owned by `rank`: contiguous blocks, sizes differing by at most one, every
    sequence owned exactly once (weak scaling: n_sequences == world gives one each)."""
    base, extra = divmod(n_sequences, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def barrier(device=None):
    """Barrier across ranks, then drain this rank's GPU (no-op pieces are skipped)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_region(fn, device=None):
    """Runs fn() bracketed by barrier + synchronize on both sides; returns the MAX elapsed
    seconds over all ranks (every rank gets the same number)."""
    import torch
    import torch.distributed as dist
    barrier(device)
    t0 = time.perf_counter()
    fn()
    barrier(device)
    dt = time.perf_counter() - t0
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_throughput(units_per_rank, seconds_max, world):
    """whole-job units/s: every rank processed `units_per_rank` in at most `seconds_max`."""
    return world * units_per_rank / seconds_max


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
