#!/bin/bash
# Round-3 session B: single-pass attention kernel (tests + A/B against the phased kernels), direction-aware training goldens,
# 176-column head slots as default, bench.py with the train / beam sections and roofline_mfma
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "tanh or attention or enc_attn or gemm_pipe" > $O/test_3b_k.log 2>&1; echo "pytest kernels rc=$?"; tail -4 $O/test_3b_k.log
for k in phased stream; do for c in 32 50 64; do
  GVD_ATTN_KERNEL=$k GVD_ATTN_CHUNK=$c timeout 120 python tools/profile_attn.py 256 10 20 2>&1 | tail -1
done; done | tee $O/attn_ab_3b.log
for k in phased stream; do for c in 50 64; do
  GVD_ATTN_KERNEL=$k GVD_ATTN_CHUNK=$c timeout 120 python tools/profile_attn.py 64 10 20 2000 5 2>&1 | tail -1
done; done | tee -a $O/attn_ab_3b.log
GVD_ATTN_KERNEL=stream timeout 120 python tools/profile_attn.py 4 10 50 2>&1 | tail -1 | tee -a $O/attn_ab_3b.log
GVD_ATTN_KERNEL=phased timeout 120 python tools/profile_attn.py 4 10 50 2>&1 | tail -1 | tee -a $O/attn_ab_3b.log
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q -x -s -p no:cacheprovider > $O/test_3b_train.log 2>&1; echo "pytest train rc=$?"; grep -i "worst\|passed\|failed\|error" $O/test_3b_train.log | tail -40
timeout 600 python bench.py > $O/bench_3b.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3b.log | cut -c1-6000
GVD_TRAIN_HEAD_PAD=192 timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_3b_192.log 2>&1; echo "bench train192 rc=$?"; tail -1 $O/bench_train_3b_192.log | cut -c1-400
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_3b_176.log 2>&1; echo "bench train176 rc=$?"; tail -1 $O/bench_train_3b_176.log | cut -c1-1200
