#!/bin/bash
# end-of-step evidence on the GPU box: full -m gpu suite, default bench line, rocprofv3 kernel stats and the two PMC
# passes of the default bench (C2).  Outputs under gpurun_out/$1
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py > $out/bench_default_$tag.json 2> $out/bench_default_$tag.err; tail -c 600 $out/bench_default_$tag.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe > $out/prof_$tag.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${tag}_$c -o solve -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-northstar --no-pipelined-probe > $out/pmc_${tag}_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_tsdf_${tag}_$c -o tsdf -- python3 $GRAFT_REPO_ROOT/tools/tsdf_kernels.py C2 5 > /dev/null 2>&1
done
find $out/prof_$tag $out/pmc_${tag}_FETCH_SIZE -name "*.csv" | head
