timeout 900 python -m pytest tests/test_gpu_solve6.py tests/test_host_cpp.py -x -q 2>&1 | tail -3
for c in C2 C3 C4; do timeout 200 python tools/ns_assemble_time.py $c 2>&1 | tail -1; done
for c in C2 C3; do python bench.py --mode northstar --config $c --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"config\"][\"workload\"][:30], d[\"value\"], d[\"ms_per_step\"])"; done
