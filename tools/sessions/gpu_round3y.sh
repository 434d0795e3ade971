#!/bin/bash
# Round-3 session Y: final state (after the dispatch changes) - chunk sweep of the attention kernel, smoke, full GPU suite, driver bench
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for c in 40 44 50 56; do GVD_ATTN_CHUNK=$c timeout 120 python tools/profile_attn.py 256 10 30 2>&1 | tail -1; done | tee $O/attn_chunk_3y.log
(timeout 300 python __graft_entry__.py smoke > $O/smoke_3y.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke_3y.log
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_tests_3y.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests_3y.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_3y.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3y.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_physical'), j['roofline']['avg_launch_us'], j['roofline_mfma']['frac'], j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'])
c = j['config']
print('b4', c['configs1_b4']); print('train', c['configs2_train_b64']['segments_per_s'], c['configs2_train_b64']['ms_per_step'], c['configs2_train_b64']['parity']['within_1e-4'], c['configs2_train_b64']['roofline']['frac'])
print('beam', c['configs4_beam5_t20_b64']['captions_per_s'], c['configs4_beam5_t20_b64']['ms_per_step'], c['configs4_beam5_t20_b64']['parity'], c['configs4_beam5_t20_b64']['roofline']['frac'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
