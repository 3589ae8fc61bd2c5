for i in 1 2; do
python bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shadow  ', d['value'], 'fps lat', d['frame_latency_ms']['median'], 'fuse', d['roofline_other'][0]['avg_launch_ms'], 'pcg', d['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --fuse-after-build --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('after-build', d['value'], 'fps lat', d['frame_latency_ms']['median'], 'fuse', d['roofline_other'][0]['avg_launch_ms'], 'pcg', d['roofline']['avg_launch_ms'])"
done
DFA_TSDF_ZCHUNK=512 python bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shadow zc512', d['value'], 'fps lat', d['frame_latency_ms']['median'], 'fuse', d['roofline_other'][0]['avg_launch_ms'], 'pcg', d['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline --no-northstar --config C1 --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C1 shadow', d['value'], 'pipelined', d['pipelined']['value'])"
python bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --config C1 --fuse-after-build --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C1 after-build', d['value'])"
