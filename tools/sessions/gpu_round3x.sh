#!/bin/bash
# Round-3 session X: 17..32-row token-loop products on the pipelined 64x64 kernel
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "lstm or gemm or top2" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_train.py -m gpu -q -x -p no:cacheprovider -k "b32 or b16 or b96 or dp8x32 or round_trip or beam" 2>&1 | tail -3
for b in 24 32; do timeout 300 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-sections 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('B=$b', j['ms_per_step'], j['value'])"; done | tee $O/b32_small_3x.log
timeout 600 python bench.py --mode train --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('train B=32', j['ms_per_step'], j['value'])" | tee -a $O/b32_small_3x.log
