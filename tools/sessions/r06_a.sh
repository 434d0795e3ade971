#!/bin/bash
# Round-6 session A: the refactored training encoder (one autograd function per layer on unpadded rows, residual gradients as
# dX addends), GVD_STRICT + the profiler test, optimiser-state pins, seed-24 tnone case; native-op profile + train bench line
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_train.py tests/test_gpu_kernels.py tests/test_gpu_stream_mm.py tests/test_gpu_train_fused.py -m gpu -q -p no:cacheprovider --timeout=600 -s > $O/r06a_tests.txt 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error" $O/r06a_tests.txt | tail -3 | cut -c1-300
grep -E "^FAILED|^ERROR" $O/r06a_tests.txt | head -30 | cut -c1-250
grep -E "torch-native device time|optimiser state after|\|update\||from the fp32 reference|step [0-9]: \|loss" $O/r06a_tests.txt | cut -c1-400
timeout 300 python tools/native_op_profile.py train > $O/r06a_native_ops_train.txt 2>&1; echo "native rc=$?"; grep -A12 "torch-native device time" $O/r06a_native_ops_train.txt | cut -c1-200
timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $O/r06a_bench_train.log 2>&1; echo "bench train rc=$?"
tail -1 $O/r06a_bench_train.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], 'lib', j.get('library_gemms'), 'mfma', j['roofline']['frac'], j['roofline']['gemm_ms_per_step'])
for r in j['roofline']['per_shape'][:40]: print(r)
" || tail -5 $O/r06a_bench_train.log
