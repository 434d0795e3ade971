#!/bin/bash
# Round-3 session U: final state - smoke, full GPU suite, the driver's bench command
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke_3u.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke_3u.log
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_tests_3u.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests_3u.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_3u.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3u.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_physical'), j['roofline']['avg_launch_us'], j['roofline_mfma']['frac'], j['parity'])
c = j['config']
print('b4', c['configs1_b4']); print('train', c['configs2_train_b64']['segments_per_s'], c['configs2_train_b64']['ms_per_step'], c['configs2_train_b64']['parity'], c['configs2_train_b64']['roofline']['frac'])
print('beam', c['configs4_beam5_t20_b64']['captions_per_s'], c['configs4_beam5_t20_b64']['ms_per_step'], c['configs4_beam5_t20_b64']['parity'], c['configs4_beam5_t20_b64']['roofline']['frac'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
