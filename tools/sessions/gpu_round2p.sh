#!/bin/bash
# Round-2 GPU session P: full per-dispatch traces of the train step and the B=4 call on HEAD
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_train /tmp/prof_b4
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -o train -- python $R/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_train.log 2>&1; echo "rocprof train rc=$?"
python $R/tools/parse_rocprof.py trace /tmp/prof_train $O/train_b64_p_trace.md "bench.py --mode train --steps 1 --warmup 1 (B=64), session P: per-dispatch groups" | head -8 | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b4 -o b4 -- python $R/bench.py --batch 4 --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_b4.log 2>&1; echo "rocprof b4 rc=$?"
python $R/tools/parse_rocprof.py trace /tmp/prof_b4 $O/bench_b4_p_trace.md "bench.py --batch 4 --steps 20 --warmup 3, session P: per-dispatch groups" | head -8 | cut -c1-200
cd $R
timeout 300 python bench.py --batch 4 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_b4.log 2>&1; echo "bench b4 rc=$?"; tail -1 $O/bench_b4.log | cut -c1-250
