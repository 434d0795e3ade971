#!/bin/bash
# att_input_mode featmap / region: the new reference goldens through the HIP path
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "lstm or bilstm" 2>&1 | tail -40 | tee $O/r05w3_bilstm.txt
