R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02e/stats -o bench -- python3 $R/bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --steps 160 --warmup 10 > $R/gpurun_out/r02e/bench_under_rocprof.json 2> $R/gpurun_out/r02e/err.txt
