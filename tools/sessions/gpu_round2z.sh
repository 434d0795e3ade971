#!/bin/bash
# Round-2 GPU session Z: fc_embed / loc_fc on the own GEMM (no library GEMM left on the inference path)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_train.py tests/test_gpu_ingest.py -m gpu -q -p no:cacheprovider > $O/test_z.log 2>&1; echo "e2e+train+ingest tests rc=$?"; tail -3 $O/test_z.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_z.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_z.log | cut -c1-330
