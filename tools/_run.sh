cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve6.py -x -q 2>&1 | tail -3
for c in C3 C2 C4; do DFA_TAG=mirror-lds python tools/ns_assemble_time.py $c 2>&1 | tail -1; done
