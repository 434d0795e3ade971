#!/bin/bash
# Round-2 GPU session L: per-dispatch kernel traces (grid sizes, idle gaps) of the train step and the decode step on HEAD
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke.log
cd /tmp; rm -rf /tmp/prof_train /tmp/prof_inf
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -o train -- python $R/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_train.log 2>&1; echo "rocprof train rc=$?"
python $R/tools/parse_rocprof.py trace /tmp/prof_train $O/train_b64_l_trace.md "bench.py --mode train --steps 1 --warmup 1 (B=64), session L: per-dispatch groups" | head -60 | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_inf -o inf -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_inf.log 2>&1; echo "rocprof inf rc=$?"
python $R/tools/parse_rocprof.py trace /tmp/prof_inf $O/bench_b256_l_trace.md "bench.py --steps 1 --warmup 1 (B=256), session L: per-dispatch groups" | head -50 | cut -c1-200
cd $R
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-300
