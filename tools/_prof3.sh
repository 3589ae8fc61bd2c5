R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02f
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f/c3 -o bench -- python3 $R/bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --config C3 --steps 40 --warmup 5 > $R/gpurun_out/r02f/c3.json 2> $R/gpurun_out/r02f/err.txt
