#!/bin/bash
# Round-4 session D: the two workgroup shapes of the inference flash kernel (8 waves x 32-key tiles, one workgroup per CU, vs
# 4 waves x 16-key tiles, two per CU): micro-benchmark dense / ragged, the headline step and the batch_size = 4 call with each
# (variant builds, tools/with_cflags.py), then the flash / encoder / greedy tests on the product build
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for sh in 8 4; do
  timeout 300 python tools/with_cflags.py flash$sh "-DGVD_FLASH_SHAPE=$sh" tools/attn_train_micro.py 0 256 2>&1 | grep -E "flash inference|with_cflags" | tee -a $O/r04d_flash_shape_ab.log
  timeout 300 python tools/with_cflags.py flash$sh "-DGVD_FLASH_SHAPE=$sh" bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sections 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('shape $sh: B=256', j['value'], 'captions/s', j['ms_per_step'], 'ms; parity', j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'], '; b4', j['config']['configs1_b4']['ms_per_call'], 'ms')" | tee -a $O/r04d_flash_shape_ab.log
done
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -p no:cacheprovider --timeout=300 -k "flash or fused_encoder or compact_preamble or greedy_matches or edge_shapes or beam_search" > $O/r04d_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r04d_tests.txt | cut -c1-300
