#!/bin/bash
# A/B of the PCG kernels on the bench configurations: tools/ab_pcg.sh "C1 C2" "default 1"
export DFA_LIB_PATH=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}/dynfu_amd/libdynfu_amd_dev.so  # the switches below exist in the development flavour only
for cfg in $1; do for v in $2; do
  if [ $v = default ]; then unset DFA_PCG_VARIANT; else export DFA_PCG_VARIANT=$v; fi
  timeout 300 python bench.py --config $cfg --steps ${3:-100} --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', '$v', d['value'], d['ms_per_step'], d['config']['pcg_iterations_last_frame'], d['solve_kernels_ms_per_frame'], d['roofline'].get('pcg_iterations_per_frame'), d['config']['max_abs_translation_error_vs_ground_truth_m'])
"
done; done
