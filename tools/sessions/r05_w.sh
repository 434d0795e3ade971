#!/bin/bash
# att_input_mode featmap / region: the new reference goldens through the HIP path
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "featmap or region or tnone or mixmul or _dp or attention or beam or bptt or attn" 2>&1 | tail -25 | tee $O/r05w2_modes.txt
