#!/bin/bash
# Round-2 GPU session N: merged-head launches of the training attention core (two-level batch in the GEMM)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "enc_attn or encoder_training or gemm" > $O/test_k.log 2>&1; echo "kernel tests rc=$?"; tail -6 $O/test_k.log
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider > $O/test_train.log 2>&1; echo "train tests rc=$?"; tail -4 $O/test_train.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-300
GVD_ENC_HEADS_MERGED=0 timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_perhead.log 2>&1; echo "bench train (per-head launches) rc=$?"; tail -1 $O/bench_train_perhead.log | cut -c1-300
