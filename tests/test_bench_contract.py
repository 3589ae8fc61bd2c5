"""CPU checks of bench.py's command-line contract (the run itself needs a GPU: `-m gpu` below)."""
import json
import os
import subprocess
import sys

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


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-frames", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["value"] > 30.0  # the north-star target
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    ns = d["northstar_mode"]  # the same frame with the 6-DoF solve, a secondary figure of the default run
    assert ns["value"] > 30.0 and ns["unit"] == "frames/s"
    sv = ns["solve"]
    assert sv["cost_per_gn"][-1] < 0.05 * sv["cost_per_gn"][0] and sv["pcgs_cut_short_by_the_launch_budget"] == 0
    assert all(0 < n < sv["pcg_iteration_cap"] for n in sv["pcg_iterations_per_gn"])  # every PCG stopped by its tolerance
    for e in [ns["roofline"]] + ns["roofline_other"]:
        for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"):
            assert key in e, key
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
