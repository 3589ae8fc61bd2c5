python -m pytest tests/test_gpu_solve.py tests/test_host_cpp.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['frame_latency_ms']['median'])"; done
for c in C1 C3 C4; do python bench.py --config $c --steps 50 --no-cpu-baseline --no-northstar --no-pipelined-probe | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'])"; done
