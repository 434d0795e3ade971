#!/bin/bash
# Round-3 session L: bi-GRU with 16 hidden units per workgroup above 64 rows (Ft=480 shape), knob matrix test
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -m gpu -q -x -p no:cacheprovider -k "gru" 2>&1 | tail -3
for hu in 8 16; do
  GVD_GRU_HU=$hu timeout 600 python bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('GRU_HU=$hu Ft=480 B=256:', j['value'], j['ms_per_step'])"
done | tee $O/gru_hu_ab_3l.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "ft480 or greedy_matches" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_knobs.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12
