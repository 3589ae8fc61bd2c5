python -m pytest tests/test_gpu_solve.py tests/test_gpu_solve6.py tests/test_gpu_warp.py tests/test_host_cpp.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline --no-northstar --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2', d['value'], 'fps lat', d['frame_latency_ms']['median'], 'fuse', d['roofline_other'][0]['avg_launch_ms'], 'pcg', d['roofline']['avg_launch_ms'], 'pipelined', d['pipelined']['value'])"
done
DFA_GRID_FOUR_KERNELS=1 python bench.py --no-cpu-baseline --no-northstar --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2 four-kernel grid', d['value'], 'fps', 'pipelined', d['pipelined']['value'])"
for c in C1 C3 C4; do
python bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --config $c --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c', d['value'], 'fps')"
done
python bench.py --no-cpu-baseline --mode northstar --config C3 --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C3 northstar', d['value'], 'fps')"
