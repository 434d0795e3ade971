#!/bin/bash
# Round-2 GPU session M2: train-path tests after the GRU dW fix
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider > $O/test_train.log 2>&1; echo "train tests rc=$?"; tail -15 $O/test_train.log
