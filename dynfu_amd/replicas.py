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


# What the single-rank self-check found (bench.py copies it into config.rccl_selfcheck): None = not attempted.
selfcheck = None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init(backend=None, device=None, single_rank_group=False):
    """Initialises the default process group when WORLD_SIZE > 1. Returns (rank, local, world).

    single_rank_group: with WORLD_SIZE absent / 1 and the nccl (= RCCL) backend, build a ONE-rank process group anyway
    (127.0.0.1, a free port, device_id) so that the N = 1 run drives the very calls the N > 1 run depends on — communicator
    creation, a device-tensor all-reduce, the barriers of the timed region — and the first multi-GPU run is not the first RCCL
    call of this code.  Nothing of it may cost the caller its measurement: every failure is caught and recorded in
    `selfcheck` = {ok, init_ms, error}, and the run goes on without a group (as a single rank always could)."""
    global selfcheck
    import torch.distributed as dist
    rank, local, world = env_world()
    if backend is None:
        backend = "nccl" if device is not None and device.type == "cuda" else "gloo"
    if world > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    elif world == 1 and single_rank_group and not dist.is_initialized() and dist.is_available():
        import datetime
        t0 = time.perf_counter()
        selfcheck = dict(ok=False, backend=backend, init_ms=None, error=None)
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            kw = dict(rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend, **kw)
            selfcheck["init_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            selfcheck["ok"] = True
        except Exception as e:  # noqa: BLE001 - a self-check must not take the run down
            selfcheck["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
            _drop_group()
    return rank, local, world


def _drop_group():
    import torch.distributed as dist
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


def _single_rank_guard(fn, what):
    """runs a collective of the one-rank self-check group; on failure records it, drops the group, returns False"""
    global selfcheck
    try:
        fn()
        return True
    except Exception as e:  # noqa: BLE001
        if selfcheck is not None:
            selfcheck["ok"] = False
            selfcheck["error"] = "%s in %s: %s" % (type(e).__name__, what, str(e)[:300])
        _drop_group()
        return False


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
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1:
            dist.barrier()
        else:  # the one-rank self-check group: the same call, guarded
            _single_rank_guard(dist.barrier, "barrier")
            if selfcheck is not None and selfcheck.get("ok"):
                selfcheck["barriers"] = selfcheck.get("barriers", 0) + 1
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_region(fn, device=None):
    """Runs fn() bracketed by barrier + synchronize on both sides; returns the MAX elapsed
    seconds over all ranks (every rank gets the same number)."""
    import torch
    import torch.distributed as dist
    single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    barrier(device)
    t0 = time.perf_counter()
    fn()
    if single:
        # one rank: there is nobody to wait for — the clock stops when this rank's GPU has drained, and the one-rank
        # self-check group runs its barrier BEHIND the measurement (an RCCL barrier costs ~1-2 ms of host time: inside the
        # window it took 15 % off a 20-step line)
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        barrier(device)
    else:
        barrier(device)
        dt = time.perf_counter() - t0
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        if dist.get_world_size() > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        elif _single_rank_guard(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX), "all_reduce(MAX)"):
            # (one rank: the maximum IS this rank's time — the reduced value is only checked, never used)
            if selfcheck is not None:
                selfcheck["max_allreduce_ok"] = bool(abs(float(t.item()) - dt) < 1e-12)
                selfcheck["ok"] = selfcheck["ok"] and selfcheck["max_allreduce_ok"]
    return dt


def aggregate_throughput(units_per_rank, seconds_max, world):
    """whole-job units/s: every rank processed `units_per_rank` in at most `seconds_max`."""
    return world * units_per_rank / seconds_max


def time_collectives(device=None, reps=5):
    """host time of the collectives the timed region pays for at N > 1, measured on whatever group exists (after the
    measurement): {"barrier_ms": [...], "allreduce_sync_ms": [...]} or None"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    out = dict(barrier_ms=[], allreduce_sync_ms=[])
    try:
        t = torch.zeros(1, dtype=torch.float64, device=device if device is not None else "cpu")
        for _ in range(reps):
            t0 = time.perf_counter()
            dist.barrier()
            out["barrier_ms"].append(round((time.perf_counter() - t0) * 1e3, 3))
        for _ in range(reps):
            t0 = time.perf_counter()
            dist.all_reduce(t)
            if device is not None and device.type == "cuda":
                torch.cuda.synchronize(device)
            out["allreduce_sync_ms"].append(round((time.perf_counter() - t0) * 1e3, 3))
    except Exception as e:  # noqa: BLE001
        out["error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
    return out


def count_ranks(device=None):
    """number of live ranks of the process group, counted by a SUM all-reduce of ones (not read from the environment);
    1 without a group"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.int64, device=device if device is not None else "cpu")
    if dist.get_world_size() > 1:
        dist.all_reduce(t)
        return int(t.item())
    if _single_rank_guard(lambda: dist.all_reduce(t), "all_reduce(SUM)") and selfcheck is not None:
        selfcheck["ranks_seen"] = int(t.item())
        selfcheck["ok"] = selfcheck["ok"] and selfcheck["ranks_seen"] == 1
    return 1


def local_slot():
    """(rank on this host, ranks on this host): LOCAL_RANK / LOCAL_WORLD_SIZE where the launcher sets them (torchrun and
    bench.py's own launcher do), the global pair otherwise.  The core slices below are per HOST: on a multi-node run the
    global rank would index past this host's core list."""
    rank, local, world = env_world()
    return (int(os.environ.get("LOCAL_RANK", rank)), int(os.environ.get("LOCAL_WORLD_SIZE", world)))


def pin_to_core_slice(rank=None, world=None):
    """In-process CPU affinity of a rank: a contiguous slice of the cores this process may use (no `taskset`: a launcher
    hop in front of a GPU program is what this pool forbids).  `rank` / `world` are the rank's slot ON THIS HOST (default:
    local_slot()).  Returns the slice or None."""
    if rank is None or world is None:
        rank, world = local_slot()
    if not 0 <= rank < world:
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per < 1:
            return None
        mine = cores[rank * per:(rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return None


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1:
            dist.destroy_process_group()
        else:
            _drop_group()
