cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve6.py -x -q 2>&1 | tail -3
python bench.py --mode northstar --config C2 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step']); [print(e['kernel'][:40], e['ms_per_frame'], e['avg_launch_ms'], e['frac']) for e in [d['roofline']]+d['roofline_other']]"
python bench.py --mode northstar --config C3 --no-cpu-baseline --steps 50 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['ms_per_step'], d['config']['last_frame']['pcg_iterations_per_gn'], d['config']['last_frame']['pcg_launches']); [print(e['kernel'][:40], e['ms_per_frame'], e['avg_launch_ms'], e['frac']) for e in [d['roofline']]+d['roofline_other']]"
