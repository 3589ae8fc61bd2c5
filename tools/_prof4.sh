R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02h
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02h/ns2 -o bench -- python3 $R/bench.py --no-cpu-baseline --mode northstar --config C2 --steps 40 --warmup 5 > $R/gpurun_out/r02h/ns2.json 2> $R/gpurun_out/r02h/err.txt
