#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
GVD_TL_EXPERIMENT=1 timeout 600 python tools/files_timeline.py 512 64 3 2 2>&1 | grep "staging while\|captions/s\|Error\|error" | tee $O/r05h_experiment.txt
