#!/bin/bash
# Round-2 GPU session F: pipelined small-M kernel (token-loop GEMMs) - tests, A/B bench, kernel statistics
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -m gpu -k "gemm or lstm or grounder" > $O/test_kernels.log 2>&1; echo "kernels rc=$?"; tail -8 $O/test_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_train.py -q -p no:cacheprovider -m gpu > $O/test_e2e.log 2>&1; echo "e2e rc=$?"; tail -6 $O/test_e2e.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-900
env GVD_GEMM_SMALL=0 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_nosmall.log 2>&1; echo "bench (general 64x64) rc=$?"; tail -1 $O/bench_nosmall.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 3 --batch 32 --no-cpu-baseline > $O/bench_b32.log 2>&1; tail -1 $O/bench_b32.log | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "rocprof stats rc=$?"
python $R/tools/parse_rocprof.py stats $O/prof_bench $O/bench_kernel_stats.md "bench.py --steps 3 --warmup 1 (B=256 greedy sample)" | sed -n 5,18p | cut -c1-160
find $O -name "*.db" -delete; find $O -name "*_trace.csv" -size +20M -delete
