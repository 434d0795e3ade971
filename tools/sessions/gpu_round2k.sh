#!/bin/bash
# Round-2 GPU session K: 4 x 1 wave split of narrow N tiles in the pipelined GEMM
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider > $O/test_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -4 $O/test_kernels.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-200
cd /tmp; rm -rf /tmp/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_train.log 2>&1; echo "rocprof train rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_train $O/train_b64_k_kernel_stats.md "bench.py --mode train --steps 2 --warmup 1 (B=64), session K" | head -16 | cut -c1-170
