#!/bin/bash
# Round-3 session Z3: device-predicated optimiser (status read after clip + Adam are enqueued): unit tests, training +
# data-parallel test files, interleaved in-process A/B of the training step, split-K sweep of the dW products
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_train_fused.py -q -p no:cacheprovider --timeout=120 > $O/fused_tests_3z3.txt 2>&1; echo "fused pytest rc=$?"; tail -5 $O/fused_tests_3z3.txt
timeout 200 python tools/train_ab.py 3 5 > $O/train_ab_3z3.log 2>&1; echo "ab rc=$?"; grep -v Detectron $O/train_ab_3z3.log | tail -5
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py -q -p no:cacheprovider --timeout=300 > $O/train_tests_3z3.txt 2>&1; echo "train pytest rc=$?"; tail -5 $O/train_tests_3z3.txt
timeout 150 python tools/dw_split_sweep.py > $O/dw_split_sweep_3z3.log 2>&1; echo "sweep rc=$?"; tail -70 $O/dw_split_sweep_3z3.log
