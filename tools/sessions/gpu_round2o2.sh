#!/bin/bash
# Round-2 GPU session O2: wave-per-sample beam-step kernel
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "beam" > $O/test_beam.log 2>&1; echo "beam tests rc=$?"; tail -4 $O/test_beam.log
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam.log | cut -c1-300
