#!/bin/bash
# rocprofv3 kernel stats + the two PMC passes of the north-star bench at C3.  Outputs under gpurun_out/*_$1
tag=${1:-ns}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o c3ns -- python3 $GRAFT_REPO_ROOT/bench.py --mode northstar --config C3 --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_$tag.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${tag}_$c -o ns -- python3 $GRAFT_REPO_ROOT/bench.py --mode northstar --config C3 --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_${tag}_$c.log 2>&1
done
tail -c 400 $out/prof_$tag.log
