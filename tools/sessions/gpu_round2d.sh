#!/bin/bash
# Round-2 GPU session D: backward GEMMs on the pipelined kernel - tests, train bench, train kernel statistics
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -m gpu -k "backward_gemms or linear_autograd or fused_encoder or gemm_nt" > $O/test_kernels.log 2>&1; echo "kernels rc=$?"; tail -12 $O/test_kernels.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py -q -p no:cacheprovider -m gpu > $O/test_train.log 2>&1; echo "train tests rc=$?"; tail -12 $O/test_train.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-500
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_train.log 2>&1; echo "rocprof train rc=$?"
python $R/tools/parse_rocprof.py stats $O/prof_train $O/train_b64_kernel_stats.md "bench.py --mode train --steps 2 --warmup 1 (B=64)" | head -40 | cut -c1-160
find $O -name "*.db" -delete; find $O -name "*_trace.csv" -size +20M -delete
