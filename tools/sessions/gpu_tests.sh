#!/bin/bash
# full GPU test suite + smoke; outputs under gpurun_out/
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"); tail -1 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/test.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/test.log
