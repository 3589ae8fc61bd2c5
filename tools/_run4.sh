mkdir -p gpurun_out
python -m pytest tests/test_gpu_tsdf.py -x -q -m gpu 2>&1 | tail -3
for zc in default 512 256 128 64; do
  if [ $zc = default ]; then unset DFA_TSDF_ZCHUNK; else export DFA_TSDF_ZCHUNK=$zc; fi
  for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2 zchunk $zc', d['value'], 'fps', d['roofline_other'][0]['avg_launch_ms'] if d.get('roofline_other') else '', d['roofline']['avg_launch_ms'])"
  done
done
unset DFA_TSDF_ZCHUNK
DFA_TSDF_LEGACY=1 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2 legacy', d['value'], 'fps', d['roofline_other'][0]['avg_launch_ms'], d['roofline']['avg_launch_ms'])"
for c in C3 C4; do
python bench.py --no-cpu-baseline --config $c --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c', d['value'], 'fps')"
DFA_TSDF_LEGACY=1 python bench.py --no-cpu-baseline --config $c --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c legacy', d['value'], 'fps')"
done
