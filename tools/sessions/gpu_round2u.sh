#!/bin/bash
# Round-2 GPU session U: phase-skewed flash attention (waves w / w+4 of a SIMD take the tile barrier at different points)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "flash or compact or encoder" > $O/test_flash.log 2>&1; echo "flash tests rc=$?"; tail -3 $O/test_flash.log
timeout 300 python tools/flash_micro.py 2>&1 | grep "^B=" | cut -c1-200
GVD_FLASH_SKEW=0 timeout 300 python tools/flash_micro.py 2>&1 | grep "^B=" | sed 's/^/[no skew] /' | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_skew.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_skew.log | cut -c1-330
GVD_FLASH_SKEW=0 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_noskew.log 2>&1; echo "bench (no skew) rc=$?"; tail -1 $O/bench_noskew.log | cut -c1-330
