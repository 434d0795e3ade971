#!/bin/bash
# Round-3 session C: where does the ~1e-3 directional gradient error of the encoder feed-forward parameters come from?
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/grad_diag.py train_masked_frame 2>&1 | grep -v Warning | tee $O/grad_diag_3c.log
