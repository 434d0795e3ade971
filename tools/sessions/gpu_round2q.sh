#!/bin/bash
# Round-2 GPU session Q: config-3 DP fixture (8 x 32 shards vs the reference), full train test file
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider > $O/test_train.log 2>&1; echo "train tests rc=$?"; tail -5 $O/test_train.log
