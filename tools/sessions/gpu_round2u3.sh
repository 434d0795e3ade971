#!/bin/bash
# Round-2 GPU session U3: flash softmax phase with lane-swap reductions + edge-only masks
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "flash or compact or encoder" > $O/test_flash.log 2>&1; echo "flash tests rc=$?"; tail -3 $O/test_flash.log
timeout 300 python tools/flash_ablate.py 2>&1 | grep "^ablate"
GVD_FLASH_SKEW=0 FA_CHILD=1 timeout 100 python tools/flash_ablate.py 2>&1 | grep "^ablate"
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "golden or greedy" > $O/test_e2e.log 2>&1; echo "e2e rc=$?"; tail -2 $O/test_e2e.log
