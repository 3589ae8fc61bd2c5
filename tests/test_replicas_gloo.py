"""world_size-2 gloo tests (CPU) of the multi-GPU path: replica assignment, the barrier-bracketed
timed region and the max-over-ranks reduction that bench.py relies on."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assign_sequences_partitions_exactly():
    from dynfu_amd.replicas import assign_sequences
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            owned = [assign_sequences(n, world, r) for r in range(world)]
            flat = [s for o in owned for s in o]
            assert flat == list(range(n))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    from dynfu_amd import replicas
    rank, local, world = replicas.init(backend="gloo")
    assert world == 2
    mine = replicas.assign_sequences(2, world, rank)
    assert mine == [rank]
    # rank 1 is the slow one: both ranks must report ITS time (max over ranks)
    def work():
        time.sleep(0.05 + 0.25 * rank)
    dt = replicas.timed_region(work)
    # the barrier really separates the ranks: a value written before it is visible after it
    t = torch.tensor([float(rank + 1)])
    torch.distributed.all_reduce(t)
    out = dict(rank=rank, dt=dt, fps=replicas.aggregate_throughput(10, dt, world), total=float(t.item()))
    print("RESULT " + json.dumps(out), flush=True)
    replicas.shutdown()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_timed_region_reports_max_over_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    import json
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
        res.append(json.loads(line[len("RESULT "):]))
    res.sort(key=lambda r: r["rank"])
    assert res[0]["dt"] == pytest.approx(res[1]["dt"])  # every rank sees the same (max) time
    assert res[0]["dt"] >= 0.29  # the slow rank's 0.30 s, not the fast rank's 0.05 s
    assert res[0]["total"] == 3.0
    assert res[0]["fps"] == pytest.approx(2 * 10 / res[0]["dt"])
