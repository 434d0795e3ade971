#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/fwd_diag.py train_masked_frame 2>&1 | grep -v "Warning\|amdgpu.ids\|detectron" | tee $O/fwd_diag_3d.log
