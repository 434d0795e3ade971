#!/bin/bash
# Round-3 session S: plain vs cooperative launches of the persistent kernels (B=4 latency)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for on in 1 0 1 0; do
  GVD_COOP_LAUNCH=$on timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('coop=$on B=4', j['ms_per_step'], j['value'])"
done | tee $O/b4_coop_3s.log
GVD_COOP_LAUNCH=0 timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "persistent or gru or b4 or stress" 2>&1 | tail -3
