#!/bin/bash
# Round-2 GPU session S2: which property of the in-situ fc7 / pool_embed calls costs the 10 % against the micro-benchmark
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for v in "" "GS_MDEV=0.8" "GS_ROWMAP=1" "GS_MDEV=0.8 GS_ROWMAP=1" "GS_RELU=1" "GS_MDEV=0.8 GS_ROWMAP=1 GS_RELU=1"; do
  env $v timeout 200 python tools/gemm_sustained.py 256256 2048 2048 1.5 2>&1 | grep -v "^trace" | tail -1
done
for v in "" "GS_MDEV=0.8"; do
  env $v timeout 200 python tools/gemm_sustained.py 256256 1024 2784 1.5 2>&1 | grep -v "^trace" | tail -1
  env $v timeout 200 python tools/gemm_sustained.py 256256 1024 1056 1.5 2>&1 | grep -v "^trace" | tail -1
  env $v timeout 200 python tools/gemm_sustained.py 256256 1024 512 1.5 2>&1 | grep -v "^trace" | tail -1
done
