#!/bin/bash
# Round-4 session I: the two reference goldens added after session H - 'MLE' at 20 frames x 100 regions (Rp = 2016 in the
# training attention core) and the four-step optimisation trajectory - on the HIP path (no library change since session H)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_e2e.py -q -p no:cacheprovider --timeout=400 -s -k "trajectory or t20" > $O/r04i_tests.txt 2>&1; echo "tests rc=$?"; grep -E "largest loss|worst|passed|failed|Error" $O/r04i_tests.txt | cut -c1-300 | tail -12
