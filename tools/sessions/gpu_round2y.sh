#!/bin/bash
# Round-2 GPU session Y (final evidence on HEAD): smoke, full GPU suite, default bench, rocprofv3 --kernel-trace --stats of the
# same command, train / beam / B=4 lines
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_tests_y.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests_y.txt
timeout 900 python bench.py > $O/bench_y.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_y.log | cut -c1-1400
cd /tmp; rm -rf /tmp/prof_y
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -o y -- python $R/bench.py --no-cpu-baseline > $O/prof_y.log 2>&1; echo "rocprof rc=$?"; tail -1 $O/prof_y.log | cut -c1-400
python $R/tools/parse_rocprof.py stats /tmp/prof_y $O/bench_b256_y_kernel_stats.md "bench.py --no-cpu-baseline (default: B=256, steps 10, warmup 2; + the B=4 section), session Y" | head -14 | cut -c1-170
cp $(find /tmp/prof_y -name '*kernel_stats.csv' | head -1) $O/bench_b256_y_kernel_stats.csv 2>/dev/null
cd $R
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > $O/bench_train_y.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train_y.log | cut -c1-400
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam_y.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam_y.log | cut -c1-300
