#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "tnone" 2>&1 | tail -40 > $O/r05d_tnone.txt; cat $O/r05d_tnone.txt
timeout 600 python bench.py --files 1024 > $O/r05d_files.log 2>&1; tail -1 $O/r05d_files.log | cut -c1-700
timeout 200 python tools/stream_mm_bench.py 64 20 softmax 2>/dev/null | cut -c1-200
