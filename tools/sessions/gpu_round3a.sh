#!/bin/bash
# Round-3 session A: tanh_fast in the attention kernels (goldens + bench + beam), the experimental K-tail / 176-slot test
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "tanh or attention or persistent or greedy or golden or beam or sample or smoke" > $O/test_3a.log 2>&1; echo "pytest rc=$?"; tail -5 $O/test_3a.log
GVD_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k 176 > $O/test_3a_176.log 2>&1; echo "pytest176 rc=$?"; tail -15 $O/test_3a_176.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_3a.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3a.log | cut -c1-1800
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam_3a.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam_3a.log | cut -c1-1500
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_3a.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train_3a.log | cut -c1-800
