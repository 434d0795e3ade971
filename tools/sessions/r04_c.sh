#!/bin/bash
# Round-4 session C: the one-head-slot K-strided GEMM (gemm_n192.hip) and the prefetching backward maps kernel: unit tests, the
# training test files, micro-benchmarks, the train bench line; post-training parity test without the Newton step (noise floor)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout=200 -k "n192 or enc_attn or encoder_training or backward_gemms" > $O/r04c_kernels.txt 2>&1; echo "kernel tests rc=$?"; tail -6 $O/r04c_kernels.txt | cut -c1-300
timeout 200 python tools/attn_train_micro.py 64 0 > $O/r04c_attn_micro.log 2>&1; echo "micro rc=$?"; grep -v Warning $O/r04c_attn_micro.log | tail -9
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider -s -k "real_optimisation" > $O/r04c_posttrain.txt 2>&1; echo "posttrain rc=$?"; grep -E "region logits|differ|passed|failed" $O/r04c_posttrain.txt | cut -c1-300
timeout 700 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_dist.py tests/test_gpu_knobs.py -q -p no:cacheprovider --timeout=400 > $O/r04c_train_tests.txt 2>&1; echo "train tests rc=$?"; tail -5 $O/r04c_train_tests.txt | cut -c1-300
timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $O/r04c_bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/r04c_bench_train.log | cut -c1-400
