R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02i
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02i/c4 -o bench -- python3 $R/bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --config C4 --steps 30 --warmup 5 > $R/gpurun_out/r02i/c4.json 2> $R/gpurun_out/r02i/err.txt
