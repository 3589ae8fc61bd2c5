for c in C2 C3 C4; do timeout 200 python tools/ns_assemble_time.py $c 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_solve6.py -x -q 2>&1 | tail -2
