#!/bin/bash
# Round-3 session J: ingest with the native .npy reader + grouped H2D copies; node-local vs unpinned reader threads, Ft = 480 and the BASELINE shape Ft = 10
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" | head -8 | tee $O/ingest_bench_3j.log
timeout 600 python tools/ingest_bench.py 256 128 8,16,32,64 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/ingest_bench_3j.log
FT=10 timeout 600 python tools/ingest_bench.py 256 128 8,16,32,64 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/ingest_bench_3j.log
timeout 600 python -m pytest tests/test_gpu_ingest.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 600 python bench.py --h2d --steps 6 --warmup 2 --no-cpu-baseline --no-sections > $O/bench_h2d_3j.log 2>&1; tail -1 $O/bench_h2d_3j.log | cut -c1-200
