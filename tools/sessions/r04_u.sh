#!/bin/bash
# Round-4 session U: bi-GRU gate phase with NU consecutive units per thread (one vector access each for the gi gates, the
# own h_{t-1} and h_t) against the previous commit's kernel (variant built from `git archive HEAD~ csrc`, GVD_VARIANT_CSRC):
# micro-benchmark + bit comparison, GRU tests, the Ft = 480 line with both builds
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=$O/r04u_gru_gates_ab.log; : > $L
P=tools/_bin/prev_src/grounded-video-description_amd/csrc
echo "[product build: vector gate accesses]" | tee -a $L
timeout 300 python tools/gru_micro.py 2>&1 | grep "^B=" | tee -a $L
echo "[previous commit]" | tee -a $L
GVD_VARIANT_CSRC=$P timeout 300 python tools/with_cflags.py gru_prev "" tools/gru_micro.py 2>&1 | grep -E "^B=" | tee -a $L
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout=300 -k "gru" 2>&1 | tail -2 | tee -a $L
for v in product prev product prev; do
  if [ $v = product ]; then C="python"; else export GVD_VARIANT_CSRC=$P; C="python tools/with_cflags.py gru_prev"; fi
  if [ $v = product ]; then unset GVD_VARIANT_CSRC; timeout 300 python bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections 2>&1 | tail -1 > /tmp/line.json
  else timeout 300 python tools/with_cflags.py gru_prev "" bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections 2>&1 | tail -1 > /tmp/line.json; fi
  python -c "
import sys, json
j = json.loads(open('/tmp/line.json').read()); print('$v: Ft=480', j['value'], j['unit'], j['ms_per_step'], 'ms')" | tee -a $L
done
