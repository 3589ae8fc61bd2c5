timeout 900 python -m pytest tests/test_gpu_solve.py -x -q 2>&1 | tail -3
lite="--no-cpu-baseline --no-northstar --no-pipelined-probe --no-live-depth --no-end-to-end"
for i in 1 2; do python bench.py $lite --steps 60 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; done
for c in C3 C4; do python bench.py --config $c $lite --steps 50 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"config\"][\"workload\"][:12], d[\"value\"], d[\"ms_per_step\"])"; done
