R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02c
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02c/stats -o bench -- python3 $R/bench.py --no-cpu-baseline --steps 160 --warmup 10 > $R/gpurun_out/r02c/bench_under_rocprof.json 2> $R/gpurun_out/r02c/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc_FETCH_SIZE -o tsdf -- python3 $R/tools/tsdf_kernels.py C2 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc_WRITE_SIZE -o tsdf -- python3 $R/tools/tsdf_kernels.py C2 5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc4_FETCH_SIZE -o tsdf -- python3 $R/tools/tsdf_kernels.py C4 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc4_WRITE_SIZE -o tsdf -- python3 $R/tools/tsdf_kernels.py C4 3 > /dev/null 2>&1
cd $R
python tools/tsdf_kernels.py C2 20 > gpurun_out/r02c/tsdf_kernels_c2.txt 2>&1
python tools/tsdf_kernels.py C4 10 > gpurun_out/r02c/tsdf_kernels_c4.txt 2>&1
python tools/tsdf_kernels.py C1 20 > gpurun_out/r02c/tsdf_kernels_c1.txt 2>&1
find gpurun_out/r02c -name "*.csv" | head -30
