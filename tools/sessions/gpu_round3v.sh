#!/bin/bash
# Round-3 session V: pipelined 64x64 kernel for the few-tile products with a device-side row count (B=4)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "greedy or b4 or edge or persistent or compact" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('B=4', j['ms_per_step'], j['value'])"; done | tee $O/b4_small_mdev_3v.log
GVD_GEMM_SMALL=0 timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('B=4 GEMM_SMALL=0', j['ms_per_step'], j['value'])" | tee -a $O/b4_small_mdev_3v.log
for b in 16 32; do timeout 300 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-sections 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('B=$b', j['ms_per_step'], j['value'])"; done | tee -a $O/b4_small_mdev_3v.log
