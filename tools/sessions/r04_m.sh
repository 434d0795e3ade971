#!/bin/bash
# Round-4 session M: the one-head-slot GEMM (gemm_n192.hip) without the per-block MFMA guards (every MFMA of the round-4
# (the two compile-time switches existed for this session only; the no-guard + prefetch form became the kernel afterwards)
# kernel was saveexec + branch + branch back: found in the ISA), and on top of that with two k tiles in flight - variant builds
# (tools/with_cflags.py): guard = the kernel as measured in sessions C - H, product = no guards, ahead = no guards + prefetch
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=$O/r04m_n192_ab.log; : > $L
timeout 200 python tools/with_cflags.py n192guard "-DGVD_N192_GUARD=1" tools/attn_train_micro.py 64 0 2>&1 | grep -E "with_cflags|train " | tee -a $L
echo "[product build]" | tee -a $L
timeout 200 python tools/attn_train_micro.py 64 0 2>&1 | grep -E "train " | tee -a $L
timeout 200 python tools/with_cflags.py n192ahead "-DGVD_N192_AHEAD=1" tools/attn_train_micro.py 64 0 2>&1 | grep -E "with_cflags|train " | tee -a $L
K="n192 or enc_attn or encoder_training or backward_gemms"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout=200 -k "$K" 2>&1 | tail -2 | tee -a $L
timeout 300 python tools/with_cflags.py n192ahead "-DGVD_N192_AHEAD=1" tools/run_pytest.py tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout=200 -k "$K" 2>&1 | tail -2 | tee -a $L
for v in product ahead; do
  if [ $v = product ]; then C="python"; else C="python tools/with_cflags.py n192ahead -DGVD_N192_AHEAD=1"; fi
  timeout 300 $C bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v: train', j['value'], j['unit'], j['ms_per_step'], 'ms; parity', json.dumps(j.get('parity', j['config'].get('parity')))[:200])" | tee -a $L
done
