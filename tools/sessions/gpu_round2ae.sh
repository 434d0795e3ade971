#!/bin/bash
# Round-2 GPU session AE: pipelined GEMM with the global loads requested two tiles ahead of their LDS write pass
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "gemm or linear or grounder or enc_attn or compact" > $O/test_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -2 $O/test_gemm.log
for shp in "256000 2048 2048" "205000 1024 2784" "205000 3168 1024" "205000 1024 1056" "205000 512 1024" "205000 1024 512"; do
  timeout 100 python tools/gemm_sustained.py $shp 0.8 2>&1 | grep -v "^trace" | tail -1 | cut -c36-200
done
