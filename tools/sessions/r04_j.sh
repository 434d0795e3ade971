#!/bin/bash
# Round-4 session J: the whole GPU suite at the final HEAD (library unchanged since session H; two goldens added since) + smoke
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r04j_gpu_tests.txt 2>&1; echo "full suite rc=$?"; tail -3 $O/r04j_gpu_tests.txt | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_train.py -q -p no:cacheprovider -s -k "trajectory" 2>&1 | grep -E "^step|largest|passed|failed" | cut -c1-200
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
