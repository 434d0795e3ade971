#!/bin/bash
# Round-3 session R: fused side kernels of the inference preamble + one-launch greedy init (B=4 latency)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "side_kernels or top2 or persistent" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ingest.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for on in 0 1 0 1; do
  GVD_SIDE_FUSED=$on timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('side_fused=$on B=4', j['ms_per_step'], j['value'])"
done | tee $O/b4_side_fused_3r.log
timeout 600 python bench.py --no-cpu-baseline --no-sections 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('B=256', j['value'], j['ms_per_step'], j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'], j['config']['configs1_b4'])" | tee -a $O/b4_side_fused_3r.log
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -p no:cacheprovider -k "decode" 2>&1 | tail -3
