#!/bin/bash
# Round-4 session R: bi-GRU with 32 hidden units per workgroup for batches of more than 128 rows (one batch tile per workgroup
# and step at B = 256) against the 16-unit form (variant build -DGVD_GRU_HU32_OFF=1, a switch that existed for this session
# only); the one-head-slot GEMM with two k tiles in flight is in both builds.  Micro-benchmark + bit comparison, the GRU /
# attention-core kernel tests, the Ft = 480 line with both builds, the train line.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=$O/r04r_gru_ab.log; : > $L
echo "[product build: HU = 32 above 128 rows]" | tee -a $L
timeout 300 python tools/gru_micro.py 2>&1 | grep "^B=" | tee -a $L
timeout 300 python tools/with_cflags.py gru_hu16 "-DGVD_GRU_HU32_OFF=1" tools/gru_micro.py 2>&1 | grep -E "with_cflags|^B=" | tee -a $L
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout=300 -k "gru or n192 or enc_attn or encoder_training or backward_gemms" 2>&1 | tail -3 | tee -a $L
for v in product hu16; do
  if [ $v = product ]; then C="python"; else C="python tools/with_cflags.py gru_hu16 -DGVD_GRU_HU32_OFF=1"; fi
  timeout 300 $C bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v: Ft=480', j['value'], j['unit'], j['ms_per_step'], 'ms; parity', json.dumps(j.get('parity'))[:160])" | tee -a $L
done
timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('product: train', j['value'], j['unit'], j['ms_per_step'], 'ms')" | tee -a $L
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider --timeout=250 -k "ft480 or greedy_matches" 2>&1 | tail -2 | tee -a $L
