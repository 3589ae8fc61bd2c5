cd $GRAFT_REPO_ROOT
for c in C1 C2 C4; do
python tools/tsdf_kernels.py $c 10 2>&1 | grep -E "^clear "
DFA_TSDF_CLEAR_LINEAR=1 python tools/tsdf_kernels.py $c 10 2>&1 | grep -E "^clear " | sed 's/^clear/clear(linear)/'
done
timeout 900 python -m pytest tests/test_gpu_tsdf.py -x -q 2>&1 | tail -3
