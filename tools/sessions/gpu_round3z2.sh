#!/bin/bash
# Round-3 session Z2: fused-kernel unit tests (tolerances fixed), rocprofv3 kernel statistics of the train step with the fused
# elementwise kernels + own Adam and with the ATen passes + torch's fused Adam
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_train_fused.py -q -p no:cacheprovider --timeout=120 > $O/fused_tests_3z2.txt 2>&1; echo "fused pytest rc=$?"; tail -5 $O/fused_tests_3z2.txt
cd /tmp
export GVD_STATS_ROWS=70
rm -rf /tmp/prof_fused
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fused -o p -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_train_fused_3z2.log 2>&1; echo "rocprof fused rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_fused $O/train_b64_z2_fused_kernel_stats.md "bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline, session Z2 (round 3): fused elementwise kernels + own clip/Adam (default)" | head -5
rm -rf /tmp/prof_aten
GVD_TRAIN_FUSED_ELEMENTWISE=0 GVD_OWN_ADAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aten -o p -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_train_aten_3z2.log 2>&1; echo "rocprof aten rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_aten $O/train_b64_z2_aten_kernel_stats.md "GVD_TRAIN_FUSED_ELEMENTWISE=0 GVD_OWN_ADAM=0 bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline, session Z2 (round 3): ATen dropout / threshold / sum passes + torch's fused Adam" | head -5
cd $R
tail -1 $O/prof_train_fused_3z2.log | cut -c1-200; tail -1 $O/prof_train_aten_3z2.log | cut -c1-200
