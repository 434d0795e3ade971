#!/bin/bash
# Round-2 GPU session M: fused training kernels (P5 row kernel fwd+bwd, GRU BPTT, in-place BPTT bookkeeping)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider > $O/test_train.log 2>&1; echo "train tests rc=$?"; tail -15 $O/test_train.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-300
GVD_P5_FUSED_TRAIN=0 GVD_GRU_TRAIN=0 timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_lib.log 2>&1; echo "bench train (library P5/GRU) rc=$?"; tail -1 $O/bench_train_lib.log | cut -c1-300
cd /tmp; rm -rf /tmp/prof_train
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -o train -- python $R/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_train.log 2>&1; echo "rocprof train rc=$?"
python $R/tools/parse_rocprof.py trace /tmp/prof_train $O/train_b64_m_trace.md "bench.py --mode train --steps 1 --warmup 1 (B=64), session M: per-dispatch groups" | head -70 | cut -c1-200
