#!/bin/bash
# Round-3 session Z1: fused elementwise kernels of the training step + own clip/Adam - unit tests, the training test file,
# train bench A/B (fused vs the ATen passes + torch's fused Adam), split-K sweep of the dW products
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_train_fused.py -q -p no:cacheprovider > $O/fused_tests_3z1.txt 2>&1; echo "fused pytest rc=$?"; tail -25 $O/fused_tests_3z1.txt
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -p no:cacheprovider > $O/train_tests_3z1.txt 2>&1; echo "train pytest rc=$?"; tail -8 $O/train_tests_3z1.txt
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_fused_3z1.log 2>&1; echo "bench fused rc=$?"; tail -1 $O/bench_train_fused_3z1.log | cut -c1-400
GVD_TRAIN_FUSED_ELEMENTWISE=0 GVD_OWN_ADAM=0 timeout 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_aten_3z1.log 2>&1; echo "bench aten rc=$?"; tail -1 $O/bench_train_aten_3z1.log | cut -c1-400
timeout 200 python tools/dw_split_sweep.py > $O/dw_split_sweep_3z1.log 2>&1; echo "sweep rc=$?"; cat $O/dw_split_sweep_3z1.log | tail -80
