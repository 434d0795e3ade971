#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_e2e.py -x -q -k "tnone" 2>&1 | tail -3
timeout 600 python tools/files_timeline.py 512 64 3 2 > $O/r05e_timeline_512.txt 2>&1; cat $O/r05e_timeline_512.txt | grep -v "^TopDown\|amdgpu.ids"
timeout 600 python tools/files_timeline.py 512 64 3 3 2>&1 | grep "captions/s"
