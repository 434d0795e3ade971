#!/bin/bash
# PREPARED at the end of round 3 (not run yet): what the compacted training layout (GVD_TRAIN_COMPACT=1) needs
# before it becomes the default - the training / data-parallel / fused-kernel test files with the knob on, the B = 64
# reference case + timing both ways, the train bench line with the knob on
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
GVD_TRAIN_COMPACT=1 timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py tests/test_gpu_train_fused.py -q -p no:cacheprovider --timeout=300 > $O/train_tests_compact_4a.txt 2>&1; echo "pytest (compact) rc=$?"; tail -8 $O/train_tests_compact_4a.txt
timeout 300 python tools/train_compact_check.py mle_b64_v5000_ft10_trained 5 2>&1 | grep -v Detectron | tee $O/train_compact_check_4a.log | tail -8
GVD_TRAIN_COMPACT=1 timeout 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_compact_4a.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_train_compact_4a.log | cut -c1-300
