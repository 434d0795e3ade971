#!/bin/bash
# Round-2 GPU session H: training encoder on the MFMA path (new kernels), A/B bench, kernel stats
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "layernorm or enc_ or encoder or gemm" > $O/test_new.log 2>&1; echo "new tests rc=$?"; tail -4 $O/test_new.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-200
GVD_ENC_TRAIN_MFMA=0 timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_lnonly.log 2>&1; echo "bench train (LN only) rc=$?"; tail -1 $O/bench_train_lnonly.log | cut -c1-200
cd /tmp
for mode in 1 0; do
  rm -rf /tmp/prof_train
  GVD_ENC_TRAIN_MFMA=$mode timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o train -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_train_$mode.log 2>&1; echo "rocprof train mfma=$mode rc=$?"
  python $R/tools/parse_rocprof.py stats /tmp/prof_train $O/train_b64_h_mfma${mode}_kernel_stats.md "bench.py --mode train --steps 2 --warmup 1 (B=64), GVD_ENC_TRAIN_MFMA=$mode" | head -30 | cut -c1-170
done
