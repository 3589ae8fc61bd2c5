cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve6.py -x -q 2>&1 | tail -4
cp dynfu_amd/libdynfu_amd.so /tmp/prod.so
DFA_TAG=waves3 python tools/ns_assemble_time.py C3 2>&1 | tail -1
DFA_TAG=waves3 python tools/ns_assemble_time.py C2 2>&1 | tail -1
for m in 4; do cp dynfu_amd/libdynfu_amd_waves$m.so.bin dynfu_amd/libdynfu_amd.so; DFA_TAG=waves$m python tools/ns_assemble_time.py C3 2>&1 | tail -1;DFA_TAG=waves$m python tools/ns_assemble_time.py C2 2>&1 | tail -1; done
cp /tmp/prod.so dynfu_amd/libdynfu_amd.so
