#!/bin/bash
# Round-2 GPU session AG: last sanity on HEAD (smoke, GEMM / flash kernel tests, greedy goldens, default bench line)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "gemm or flash or lstm" > $O/test_ag.log 2>&1; echo "kernel subset rc=$?"; tail -1 $O/test_ag.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "greedy_matches_reference" > $O/test_ag2.log 2>&1; echo "greedy goldens rc=$?"; tail -1 $O/test_ag2.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_ag.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_ag.log | cut -c1-300
