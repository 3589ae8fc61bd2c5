#!/bin/bash
# A/B of the frame schedule: tools/ab_fuse.sh "C1 C2" [extra bench flags]
for cfg in $1; do for v in "" "--fuse-first"; do
  timeout 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline $v $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', '$v' or 'fuse-behind-graph', d['value'], d['ms_per_step'], d.get('solve_kernels_ms_per_frame'), [ (r['kernel'][:20], r['avg_launch_ms']) for r in [d['roofline']]+d['roofline_other'] if 'avg_launch_ms' in r])
"
done; done
