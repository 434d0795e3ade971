#!/bin/bash
# Round-6 session B: (1) tests of the fixes of session A + the one-exponential beam score; (2) maps-kernel start skew sweep;
# (3) beam attention kernel alone; (4) N = 1056 / 448 GEMM shapes with the edge-tile skip at any tile count
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_strict.py tests/test_gpu_train.py tests/test_gpu_stream_mm.py -m gpu -q -p no:cacheprovider --timeout=600 -s > $O/r06b_tests1.txt 2>&1; echo "tests1 rc=$?"
grep -E "passed|failed" $O/r06b_tests1.txt | tail -2 | cut -c1-300; grep -E "^FAILED|^ERROR" $O/r06b_tests1.txt | head -20 | cut -c1-250
grep -E "torch-native device time|optimiser state after|\|update\||from the fp32 reference|step [0-9]: \|loss" $O/r06b_tests1.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider --timeout=600 -s -k "attention or beam or enc or encoder or tanh" > $O/r06b_tests2.txt 2>&1; echo "tests2 rc=$?"
grep -E "passed|failed" $O/r06b_tests2.txt | tail -2 | cut -c1-300; grep -E "^FAILED|^ERROR" $O/r06b_tests2.txt | head -20 | cut -c1-250
grep -E "region scores vs fp64" $O/r06b_tests2.txt | head -8
for k in 0 2 4 6 9 12; do echo "== GVD_MAPS_SKEW=$k"; GVD_MAPS_SKEW=$k timeout 200 python tools/attn_train_micro.py 64 0 2>&1 | grep -E "bwd maps|core forward"; done | tee $O/r06b_maps_skew.txt
for i in 1 2; do timeout 200 python tools/profile_attn.py 64 10 20 2000 5 2>&1 | tail -1; done | tee $O/r06b_beam_attn.txt
for e in 0 1; do GVD_GEMM_EDGE=$e timeout 200 python tools/gemm_shape_micro.py dx:64000,1024,1056 dw:64000,1024,1056 fwd:64000,448,2048 dw:64000,448,2048 dx:64000,1024,3168 fwd:64000,3168,1024 dx:64000,1024,2784 2>&1 | grep -v Warn; done | tee $O/r06b_gemm_edge.txt
