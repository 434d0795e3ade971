#!/bin/bash
# Round-2 GPU session G: full GPU suite on HEAD, default bench, overlap variant, beam and train lines
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/test.log 2>&1; echo "pytest rc=$?"; tail -8 $O/test.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-1300
timeout 600 python bench.py --steps 10 --warmup 2 --overlap --no-cpu-baseline > $O/bench_overlap.log 2>&1; echo "bench overlap rc=$?"; tail -1 $O/bench_overlap.log | cut -c1-1300
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam.log | cut -c1-300
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-300
