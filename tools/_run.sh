cd $GRAFT_REPO_ROOT
for rc in 192 256 320; do DFA_S6_RC=$rc DFA_TAG=rc$rc python tools/ns_assemble_time.py C3 2>&1 | tail -1; done
DFA_TAG=k4 python tools/ns_assemble_time.py C2 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_solve6.py -x -q 2>&1 | tail -3
