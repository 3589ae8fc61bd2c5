cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve6.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns -o k -- python3 $GRAFT_REPO_ROOT/bench.py --mode northstar --config C3 --no-cpu-baseline --steps 20 > /tmp/prof_ns.log 2>&1
grep '^{' /tmp/prof_ns.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['ms_per_step'])"
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_ns/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("  %-64s calls %6s avg %9.2f us  %5s%%" % (r['Name'].replace('(anonymous namespace)::','')[:64], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
