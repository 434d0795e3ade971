#!/bin/bash
# Round-2 GPU session W: class-logit GEMM over 448 padded classes (vectorised epilogue)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x > $O/test_ke.log 2>&1; echo "kernel+e2e tests rc=$?"; tail -3 $O/test_ke.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_w.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_w.log | cut -c1-330
