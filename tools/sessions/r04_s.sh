#!/bin/bash
# Round-4 session S: bi-GRU with one-level barriers per (direction, batch-tile group) sub-grid instead of the grid-wide tree
# (variant build -DGVD_GRU_SUBGRID_OFF=1 = the tree, a switch for this session only): micro-benchmark + bit comparison, GRU
# tests, timeout knob modes, Ft = 480 and headline lines with both builds
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=$O/r04s_gru_subgrid_ab.log; : > $L
echo "[product build: sub-grid barriers]" | tee -a $L
timeout 300 python tools/gru_micro.py 2>&1 | grep "^B=" | tee -a $L
timeout 300 python tools/with_cflags.py gru_tree "-DGVD_GRU_SUBGRID_OFF=1" tools/gru_micro.py 2>&1 | grep -E "with_cflags|^B=" | tee -a $L
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout=300 -k "gru" 2>&1 | tail -2 | tee -a $L
timeout 400 python -m pytest tests/test_gpu_knobs.py -q -p no:cacheprovider --timeout=300 2>&1 | tail -2 | tee -a $L
for v in product tree; do
  if [ $v = product ]; then C="python"; else C="python tools/with_cflags.py gru_tree -DGVD_GRU_SUBGRID_OFF=1"; fi
  timeout 300 $C bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v: Ft=480', j['value'], j['unit'], j['ms_per_step'], 'ms')" | tee -a $L
done
timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_train.py -q -p no:cacheprovider --timeout=250 -k "ft480 or greedy_matches or mle_gradients" 2>&1 | tail -2 | tee -a $L
