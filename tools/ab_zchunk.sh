#!/bin/bash
# frame rate against the z-chunk of the TSDF sweep (fewer, longer workgroups leave room for the solve's kernels)
export DFA_LIB_PATH=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}/dynfu_amd/libdynfu_amd_dev.so  # the switches below exist in the development flavour only
for cfg in $1; do for v in 0 64 128 256 512; do
  DFA_TSDF_ZCHUNK=$v timeout 300 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', 'zchunk $v', d['value'], d['ms_per_step'], d.get('solve_kernels_ms_per_frame'), d['frame_latency_ms']['median'], [r['avg_launch_ms'] for r in d['roofline_other']])
"
done; done
