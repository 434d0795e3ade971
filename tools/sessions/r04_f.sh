#!/bin/bash
# Round-4 session F: ragged flash attention with the device-built tile map (live workgroups first): tests, micro-benchmark,
# headline step
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -p no:cacheprovider --timeout=300 -k "flash or fused_encoder or compact_preamble or greedy_matches or edge_shapes or beam_search or row_map" > $O/r04f_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r04f_tests.txt | cut -c1-300
timeout 200 python tools/attn_train_micro.py 0 256 > $O/r04f_attn_micro.log 2>&1; echo "micro rc=$?"; grep -v Warning $O/r04f_attn_micro.log | tail -5
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sections > $O/r04f_bench.log 2>&1; tail -1 $O/r04f_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('B=256', j['value'], 'captions/s', j['ms_per_step'], 'ms; parity', j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'], '; b4', j['config']['configs1_b4']['ms_per_call'], 'ms; attn', j['roofline']['avg_launch_us'], j['roofline']['frac'])"
