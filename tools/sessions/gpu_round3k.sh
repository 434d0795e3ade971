#!/bin/bash
# Round-3 session K: split-K for the few-row frame-side products (B=4 latency)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "split_k or gemm or gru" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for on in 0 1 0 1; do
  GVD_SPLIT_K=$on timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('split_k=$on', j['ms_per_step'], j['value'])"
done | tee $O/b4_splitk_fewrow_3k.log
