cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve6.py -x -q 2>&1 | tail -6
for f in adaptive geometric; do
python bench.py --mode northstar --config C2 --no-cpu-baseline --forcing $f | python -c "
import json,sys; d=json.loads(sys.stdin.read()); lf=d['config']['last_frame']; print('C2 $f', d['value'], d['ms_per_step'], lf['pcg_iterations_per_gn'], lf['pcg_tolerance_per_gn'], lf['pcg_launches'], lf['cost_per_gn'])"
python bench.py --mode northstar --config C3 --no-cpu-baseline --steps 50 --forcing $f | python -c "
import json,sys; d=json.loads(sys.stdin.read()); lf=d['config']['last_frame']; print('C3 $f', d['value'], d['ms_per_step'], lf['pcg_iterations_per_gn'], lf['pcg_launches'], lf['cost_per_gn'])"
done
