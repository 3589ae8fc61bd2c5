#!/bin/bash
# bench lines of the configurations that are not the default one (reference mode and north-star mode), the live-depth
# workload and the default line itself -> gpurun_out/$1_configs.jsonl
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_configs.jsonl
: > $out
lite="--no-cpu-baseline --no-northstar --no-pipelined-probe --no-live-depth --no-end-to-end"
for c in C1 C3 C4; do python bench.py --config $c $lite --steps 50 2>/dev/null | tail -1 >> $out; done
for c in C1 C2 C3 C4; do python bench.py --mode northstar --config $c --no-cpu-baseline --steps 40 2>/dev/null | tail -1 >> $out; done
python bench.py --live depth --steps 60 2>/dev/null | tail -1 >> $out
python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_default.json
wc -l $out; python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln); print(d["config"]["workload"][:60], d["value"], d["ms_per_step"])
d = json.load(open("gpurun_out/${tag}_default.json")); print("default", d["value"], d["northstar_mode"]["value"], d["live_depth_mode"]["value"], d["end_to_end"]["ref"]["median_ms"], d["end_to_end"]["northstar"]["median_ms"], d["cpu_baseline"]["value"])
PY
