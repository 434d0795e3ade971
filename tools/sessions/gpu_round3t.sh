#!/bin/bash
# Round-3 session T: plain launches of the persistent kernels as the default - persistent / GRU / e2e / knob tests + default bench
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py tests/test_gpu_train.py -m gpu -q -x -p no:cacheprovider -k "persistent or gru or b4 or stress or golden or greedy" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_knobs.py -m gpu -q -p no:cacheprovider -k "decode" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-sections 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('B=256', j['value'], j['ms_per_step'], j['parity']['token_ids_equal'], j['config']['configs1_b4'])"
