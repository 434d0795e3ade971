#!/bin/bash
# Round-2 GPU session AI: direct-to-LDS operand loads in the pipelined GEMM as the default - kernel tests, goldens, bench
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "gemm or linear or grounder or enc_attn or compact" > $O/test_ai.log 2>&1; echo "gemm-path kernel tests rc=$?"; tail -1 $O/test_ai.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "greedy_matches_reference or beam_search_matches_reference" > $O/test_ai2.log 2>&1; echo "goldens rc=$?"; tail -1 $O/test_ai2.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_ai.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_ai.log | cut -c1-300
timeout 120 python -m pytest tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -k "mle_gradients or fused_paths or one_step" > $O/test_ai3.log 2>&1; echo "train subset rc=$?"; tail -1 $O/test_ai3.log
