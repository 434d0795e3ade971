#!/bin/bash
# Round-2 GPU session AB: sanity after the per-device pack-cache key (smoke + greedy / beam goldens + default bench line)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "greedy or beam or pipelined" > $O/test_ab.log 2>&1; echo "e2e subset rc=$?"; tail -2 $O/test_ab.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_ab.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_ab.log | cut -c1-330
