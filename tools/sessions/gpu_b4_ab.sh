#!/bin/bash
# A/B runs of the batch-4 path under the tuning knobs + the kernel / e2e parity tests
set -u
run() { echo "== $*"; env "$@" timeout 200 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -5
run GVD_GEMV_KS=1
run GVD_GEMV_KS=2
run GVD_GEMV_KS=4
run GVD_GEMV_KS=4 GVD_GEMM_VARIANT=0
