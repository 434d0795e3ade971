#!/bin/bash
# Round-2 GPU session J: compacted features consumed in place (attention row map, fc7 fused row gather), flash dead-wave skip
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x > $O/test_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -4 $O/test_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x > $O/test_e2e.log 2>&1; echo "e2e tests rc=$?"; tail -4 $O/test_e2e.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-1500
GVD_FC7_ROWMAP=0 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_nofc7map.log 2>&1; echo "bench (fc7 gather) rc=$?"; tail -1 $O/bench_nofc7map.log | cut -c1-200
