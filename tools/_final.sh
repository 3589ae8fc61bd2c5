R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02g
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r02g/bench_driver_like.json 2> $R/gpurun_out/r02g/bench_driver_like.err
python bench.py > $R/gpurun_out/r02g/bench_default.json 2> $R/gpurun_out/r02g/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02g/stats -o bench -- python3 $R/bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe > $R/gpurun_out/r02g/bench_under_rocprof.json 2> $R/gpurun_out/r02g/err.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02g/pmc_FETCH_SIZE -o solve -- python3 $R/bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02g/pmc_WRITE_SIZE -o solve -- python3 $R/bench.py --no-cpu-baseline --no-northstar --no-pipelined-probe --steps 10 --warmup 2 > /dev/null 2>&1
ls $R/gpurun_out/r02g
