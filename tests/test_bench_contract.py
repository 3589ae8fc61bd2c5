"""CPU checks of bench.py's command-line contract (the run itself needs a GPU: `-m gpu` below)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_defaults_and_flags():
    sys.path.insert(0, ROOT)
    import bench
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0 and a.config == "C2" and a.mode == "ref"
    sys.argv = ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup) == (4, 7, 2)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


DETAIL = os.path.join(tempfile.gettempdir(), "dfa_bench_detail_%d.json" % os.getpid())


def _bench(*argv, launcher=None, timeout=300):
    if os.path.exists(DETAIL):
        os.remove(DETAIL)
    cmd = [sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py")] + list(argv) + ["--detail-file", DETAIL]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR",
                                                             "MASTER_PORT", "DFA_BENCH_LAUNCHED_BY")}
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


def _strings(x):
    if isinstance(x, str):
        yield x
    elif isinstance(x, dict):
        for k, v in x.items():
            yield k
            yield from _strings(v)
    elif isinstance(x, (list, tuple)):
        for v in x:
            yield from _strings(v)


def _line(stdout):
    """the ONE stdout line, checked against the limits the driver's reader needs (r05: a 28 KB line came back unparsed)"""
    lines = stdout.strip().splitlines()
    assert len(lines) == 1, stdout[-1500:]
    assert len(lines[0].encode()) < 4096, len(lines[0])
    d = json.loads(lines[0])
    assert max(len(t) for t in _strings(d)) <= 160
    return d


def _detail():
    with open(DETAIL) as f:
        return json.load(f)


def test_contract_line_of_a_fat_record_stays_under_4_kb():
    """the line builder on the largest record this repository has produced (round 5's 28 KB line, kept under profiles/)
    and on a record whose every secondary figure is present: under 4 KB, no long strings, the contract's fields intact"""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        fat = json.load(f)
    assert len(json.dumps(fat)) > 20000
    fat["config"]["repeats"] = 5
    for extra in ({}, {"multi_sequence": {str(i): {"value": 1234.56 + i} for i in range(400)}}):
        text = bench.contract_line(dict(fat, **extra), "bench_detail.json")
        assert "\n" not in text and len(text.encode()) < bench.LINE_LIMIT
        d = json.loads(text)
        assert max(len(t) for t in _strings(d)) <= bench.STRING_LIMIT
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail_file"):
            assert key in d, key
        assert d["value"] == fat["value"] and d["config"]["workload"] and d["config"]["rccl_selfcheck"]["ok"] is True
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in d["roofline"], key
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] == fat["cpu_baseline"]["value"] and cb["single_thread"] > 0 and cb["all_cores"] > 0
    assert d.get("secondary") is None or "multi_sequence" not in d["secondary"]  # the oversized part was what went
    text = bench.contract_line(fat, "bench_detail.json")
    sec = json.loads(text)["secondary"]
    assert sec["other_configs"]["C3_ref"] == fat["other_configs"]["C3_ref"]["value"]
    assert set(sec["other_configs"]) >= {"C1_ref", "C3_ref", "C3_northstar", "C4_ref", "C4_northstar"}


def test_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` outside torchrun: the launcher starts two rank processes (gloo, no GPU work), the line
    comes from two LIVE ranks (counted by an all-reduce) — SURVEY 8(e), BASELINE config 5."""
    r = _bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "5", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)  # one line, rank 0's
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["steps"] == 5 and d["scaling"] == "weak"
    assert d["config"]["launcher"] == "bench.py" and d["config"]["repeats"] == 5 and len(_detail()["region_ms"]) == 5
    # 5 steps of 20 ms on each of two ranks in the time of one: whole-job value = 2 x 5 / max time
    assert d["value"] == pytest.approx(2 * 5 / (d["ms_per_step"] * 5e-3), rel=1e-3)
    assert 40.0 < d["value"] < 100.5  # (a loaded host stretches the sleeps; never faster than the sleeps allow)


def test_launcher_fails_when_a_rank_dies():
    r = _bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "5", "--warmup", "1", "--dry-run-fail-rank", "1")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no line from a run that lost a rank
    assert "rank 1 of 2" in r.stderr


def test_launcher_gives_up_on_a_rank_that_hangs():
    """a rank that neither fails nor finishes: the launcher ends the run at its wall-clock limit instead of waiting for ever"""
    r = _bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0", "--dry-run-hang-rank", "1",
               "--rank-timeout", "8", timeout=120)
    assert r.returncode != 0 and "--rank-timeout" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_rank_run_builds_a_one_rank_process_group():
    """N = 1 outside torchrun: a one-rank process group is built anyway so that the same init / all-reduce / barrier calls
    the N > 1 run depends on have run (gloo here; nccl = RCCL on the GPU box: config.rccl_selfcheck of the bench line)."""
    r = _bench("--gpus", "1", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["config"]["rccl_selfcheck"]["ok"] is True and d["config"]["launcher"] is None
    sc = _detail()["config"]["rccl_selfcheck"]
    assert sc["ok"] is True and sc["error"] is None and sc["ranks_seen"] == 1 and sc["barriers"] >= 2 and sc["max_allreduce_ok"] is True
    assert d["n_gpus"] == 1
    r = _bench("--gpus", "1", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0", "--no-rccl-selfcheck")
    assert r.returncode == 0 and "skipped" in _line(r.stdout)["config"]["rccl_selfcheck"]


def test_forced_launcher_with_one_rank():
    """`--gpus 1 --force-launcher`: the N = 1 run through the rank launcher (fresh child, WORLD_SIZE=1, core slice, rank 0's
    line forwarded) — the path every N > 1 run takes (gloo here; `-m gpu`: the same on the GPU box with RCCL)"""
    r = _bench("--gpus", "1", "--force-launcher", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["ranks_seen"] == 1 and d["config"]["launcher"] == "bench.py"
    assert d["config"]["rccl_selfcheck"]["ok"] is True


def test_self_check_failure_costs_the_line_nothing():
    """the one-rank group cannot be built (its rendezvous port is taken): recorded, and the run goes on without a group"""
    import socket
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0",
           "--detail-file", DETAIL]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    busy = socket.socket()
    busy.bind(("127.0.0.1", 0))
    busy.listen(1)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(busy.getsockname()[1]))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    finally:
        busy.close()
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["config"]["rccl_selfcheck"]["ok"] is False and _detail()["config"]["rccl_selfcheck"]["error"] and d["n_gpus"] == 1


def test_ranks_pin_themselves_to_disjoint_core_slices():
    code = ("import os, sys; sys.path.insert(0, %r); from dynfu_amd import replicas; "
            "print(sorted(replicas.pin_to_core_slice(int(sys.argv[1]), 2) or []), sorted(os.sched_getaffinity(0)))" % ROOT)
    outs = [subprocess.run([sys.executable, "-c", code, str(r)], capture_output=True, text=True, timeout=120) for r in range(2)]
    assert all(o.returncode == 0 for o in outs), outs[0].stderr[-500:]
    a, b = (eval(o.stdout.strip().replace("] [", "],["))[0] for o in outs)
    if len(os.sched_getaffinity(0)) >= 2:
        assert a and b and not set(a) & set(b) and max(a) < min(b)
        assert a == list(range(a[0], a[0] + len(a))) or len(a) == len(set(a))  # contiguous in the allowed set's order


def test_same_line_under_torchrun():
    """the driver's N > 1 command: torch.distributed.run starts the ranks, bench.py must not start more"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "0",
               launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port)])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["launcher"] == "torchrun"


def test_world_size_must_match_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run", "--backend", "gloo", "--detail-file", DETAIL],
                       capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    r = _bench("--steps", "5", "--warmup", "2", "--cpu-frames", "1", timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    # ONE line on stdout, the JSON one, under 4 KB with no long strings — RCCL's version block and every other library's
    # chatter go to stderr (claim_stdout); everything beyond the contract's fields is in the detail file
    line = _line(r.stdout)
    d = _detail()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line and key in d, key
        if key not in ("config", "roofline", "cpu_baseline"):
            assert line[key] == d[key], key
    assert line["detail_file"]
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in line["config"] and d["value"] > 30.0  # the north-star target
    # value / ms_per_step are those of the median of `repeats` back-to-back regions over the same frames
    assert line["config"]["repeats"] == 5 and len(d["region_ms"]) == 5
    assert sorted(d["region_ms"])[2] == pytest.approx(d["ms_per_step"] * d["steps"], rel=1e-3)
    # N = 1 drives a one-rank RCCL process group: init with device_id, a device-tensor all-reduce, the timed region's barriers
    sc = d["config"]["rccl_selfcheck"]
    assert sc["ok"] is True and sc["backend"] == "nccl" and sc["ranks_seen"] == 1 and sc["barriers"] >= 2, sc
    assert line["config"]["rccl_selfcheck"]["ok"] is True and line["config"]["rccl_selfcheck"]["init_ms"] > 0
    # the roofline of the dominant kernel: the HBM view (algorithmic bytes / launch time / 8 TB/s), in the line and in the file
    for rf in (line["roofline"], d["roofline"]):
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_frame",
                    "algorithmic_bytes_per_launch"):
            assert key in rf, key
        assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
        assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9, rel=0.02)
    assert line["roofline"]["traffic_over_algorithmic"] is None or line["roofline"]["traffic_over_algorithmic"] > 0
    sec = line["secondary"]
    for key in ("C1_ref", "C3_ref", "C3_northstar", "C4_ref", "C4_northstar"):
        assert sec["other_configs"][key] == pytest.approx(d["other_configs"][key]["value"], rel=1e-3)
    assert sec["northstar_mode"] == pytest.approx(d["northstar_mode"]["value"], rel=1e-3)
    lcb = line["cpu_baseline"]
    assert lcb["kind"] == "port" and lcb["value"] == d["cpu_baseline"]["value"] and lcb["cores"] == d["cpu_baseline"]["cores"]
    assert lcb["single_thread"] > 0 and lcb["all_cores"] > 0 and lcb["host_cores"] >= 1 and lcb["sample"]
    ns = d["northstar_mode"]  # the same frame with the 6-DoF solve, a secondary figure of the default run
    assert ns["value"] > 30.0 and ns["unit"] == "frames/s"
    sv = ns["solve"]
    assert sv["cost_per_gn"][-1] < 0.05 * sv["cost_per_gn"][0] and sv["pcgs_cut_short_by_the_launch_budget"] == 0
    assert all(0 < n < sv["pcg_iteration_cap"] for n in sv["pcg_iterations_per_gn"])  # every PCG stopped by its tolerance
    # the Gauss-Newton stopping rule (dfa_solve6_params.gn_tol, the reference's earlyOut): inside an outer iteration no
    # accepted linearisation has a higher energy than the one before it (beyond gn_tol); the fixed-iteration run is beside it
    assert sv["gn_tol"] > 0 and sv["gn_solves"] <= sv["gn_iterations"]
    for costs in sv["cost_per_gn_by_outer_iteration"]:
        assert all(b <= a * (1 + sv["gn_tol"]) * (1 + 1e-4) for a, b in zip(costs, costs[1:])), costs
    fx = ns["fixed_iterations"]
    assert fx["gn_solves"] >= sv["gn_solves"] and fx["value"] > 30.0 and sv["final_cost"] <= fx["final_cost"] * 1.02
    for e in [ns["roofline"]] + ns["roofline_other"]:
        for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"):
            assert key in e, key
        # SURVEY 8(d)'s bytes are what `frac` is computed from; the design's own data flow is kept beside them
        assert e["survey_bytes_per_launch"] > 0 and e["implementation_bytes_per_launch"] >= e["survey_bytes_per_launch"] * 0.99
        assert abs(e["frac"] - e["survey_bytes_per_launch"] / (e["avg_launch_ms"] * 1e-3) / 1e9 / e["peak"]) < 2e-3
    cb6 = ns["cpu_baseline"]  # the CPU statement of the north-star frame beside the north-star figure
    assert cb6["kind"] == "port" and cb6["cores"] >= 1 and cb6["value"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    # the thread count is swept, not guessed: `value` is the best of the sweep, `cores` its thread count
    assert cb["by_threads"] and cb["value"] == max(v["value"] for v in cb["by_threads"].values())
    assert str(cb["cores"]) in cb["by_threads"]
    # single thread and all host cores (nproc stated) beside it
    assert cb["single_thread"]["cores"] == 1 and cb["single_thread"]["value"] > 0
    # "all cores" = every CPU the process can keep busy: the cgroup quota where the box sets one (16 of 256 on this pool)
    hc = cb["host_cores"]
    assert cb["all_cores"]["cores"] == hc["usable"] >= 1 and cb["all_cores"]["value"] > 0
    assert hc["usable"] <= hc["nproc"] and (hc["cgroup_quota_cpus"] is None or hc["usable"] <= hc["cgroup_quota_cpus"] + 0.5)
    # the oracle's translations of the CPU sample's last frame against the HIP solve of the same frame
    assert d["config"]["max_abs_translation_diff_vs_oracle_m"] <= 2e-5
    c1 = d["other_configs"]["C1_ref"]  # BASELINE config 1 IS the CPU-path configuration: timed on both sides
    assert c1["value"] > 30.0 and c1["cpu_baseline"]["value"] > 0 and c1["cpu_baseline"]["single_thread"]["cores"] == 1
    # the raycast, reported separately, priced by the work of its own rays
    rc = d["raycast"]
    assert rc["work"]["hits"] > 0.5 * 640 * 480 and rc["work"]["march_fetches"] > 10 * rc["work"]["hits"]
    for variant in ("points", "depth"):
        e = rc[variant]
        assert 0 < e["avg_launch_ms"] < 1.0 and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-3
        assert e["unique_voxel_bytes_per_launch"] < e["survey_bytes_per_launch"]
    # the other BASELINE configurations under the same clock, both modes
    oc = d["other_configs"]
    for key in ("C3_ref", "C3_northstar", "C4_ref", "C4_northstar"):
        assert "error" not in oc[key], oc[key]
        assert oc[key]["value"] > 30.0 and oc[key]["unit"] == "frames/s"
    assert oc["C3_northstar"]["solve"]["pcgs_cut_short_by_the_launch_budget"] == 0


@pytest.mark.gpu
def test_northstar_line_cold_and_warm_start():
    """`--mode northstar` prints the same contract line; `--warm-start` (a frame starts from the transforms the frame before
    solved) runs the same frames, converges, and is not slower than starting every frame from the canonical state."""
    vals = {}
    for name, extra in (("cold", []), ("warm", ["--warm-start"])):
        r = _bench("--mode", "northstar", "--config", "C2", "--steps", "12", "--warmup", "4", "--no-cpu-baseline",
                   "--no-rccl-selfcheck", *extra, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = _line(r.stdout)
        d = _detail()
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "config"):
            assert key in d and key in line, key
        assert line["value"] == d["value"] and line["roofline"]["bound"] == "hbm"
        lf = d["config"]["last_frame"]
        assert d["steps"] == 12 and d["value"] > 30.0 and lf["final_cost"] < 0.1 * max(lf["cost_per_gn"][0], 1e-3) + 0.1
        vals[name] = (d["value"], lf)
    assert vals["warm"][0] > 0.9 * vals["cold"][0]
    assert vals["warm"][1]["cost_per_gn"][0] < vals["cold"][1]["cost_per_gn"][0]  # (it starts nearer to the frame's surface)


@pytest.mark.gpu
def test_forced_launcher_on_the_gpu_box():
    """the rank launcher (launch_ranks: a fresh child per rank, its core slice, rank 0's stdout forwarded,
    the wall-clock limit) had never run on a GPU box — N = 1 bypasses it.  `--force-launcher` takes the N = 1 run through it
    with WORLD_SIZE=1 and RCCL: one line, one live rank, the self-check green."""
    r = _bench("--gpus", "1", "--force-launcher", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-end-to-end",
               "--no-other-configs", "--no-northstar", "--no-live-depth", "--no-multi-sequence", "--no-raycast",
               "--no-pipelined-probe", "--rank-timeout", "600", timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["ranks_seen"] == 1 and line["config"]["launcher"] == "bench.py"
    assert line["config"]["rccl_selfcheck"]["ok"] is True and line["config"]["rccl_selfcheck"]["backend"] == "nccl"
    assert line["value"] > 30.0 and line["steps"] == 5


@pytest.mark.gpu
def test_c3_line_prices_the_team_pcg():
    """`--config C3` (4 096 nodes: above the register-resident kernels): the dominant kernel of the line is the team PCG
    (pcg_team_kernel: three teams of persistent workgroups, a coordinate per XCD), priced on the HBM view like every other; no
    team gave up during the run."""
    r = _bench("--config", "C3", "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-end-to-end", "--no-other-configs",
               "--no-northstar", "--no-live-depth", "--no-multi-sequence", "--no-raycast", "--no-pipelined-probe", timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line, d = _line(r.stdout), _detail()
    rf = d["roofline"]
    assert "pcg_team_kernel" in rf["kernel"] and line["roofline"]["kernel"].startswith("pcg_team_kernel")
    assert rf["team_pcg"]["aborts"] == 0 and not rf["team_pcg"]["disabled"] and rf["team_pcg"]["launches"] > 0
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches_per_frame"] == 10
    assert line["value"] > 30.0 and line["config"]["max_abs_translation_error_vs_ground_truth_m"] < 1e-4
