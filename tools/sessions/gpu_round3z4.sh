#!/bin/bash
# Round-3 session Z4 (HEAD after the training-tail work): fused-kernel unit tests, training + data-parallel test files,
# interleaved in-process A/B of the training step, the driver's bench command, rocprofv3 kernel statistics of the train step
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_train_fused.py -q -p no:cacheprovider --timeout=120 > $O/fused_tests_3z4.txt 2>&1; echo "fused pytest rc=$?"; tail -4 $O/fused_tests_3z4.txt
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py -q -p no:cacheprovider --timeout=300 > $O/train_tests_3z4.txt 2>&1; echo "train pytest rc=$?"; tail -3 $O/train_tests_3z4.txt
timeout 200 python tools/train_ab.py 3 5 > $O/train_ab_3z4.log 2>&1; echo "ab rc=$?"; grep "ms/step" $O/train_ab_3z4.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_3z4.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3z4.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_physical'), j['roofline']['avg_launch_us'], j['roofline_mfma']['frac'], j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'])
c = j['config']
print('b4', c['configs1_b4']); print('train', c['configs2_train_b64']['segments_per_s'], c['configs2_train_b64']['ms_per_step'], c['configs2_train_b64']['parity']['within_1e-4'], c['configs2_train_b64']['roofline']['frac'])
print('beam', c['configs4_beam5_t20_b64']['captions_per_s'], c['configs4_beam5_t20_b64']['ms_per_step'], c['configs4_beam5_t20_b64']['parity'], c['configs4_beam5_t20_b64']['roofline']['frac'])
print('ft480', c['ft480_b256'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
cd /tmp; export GVD_STATS_ROWS=70; rm -rf /tmp/prof_train
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/prof_train_3z4.log 2>&1; echo "rocprof rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_train $O/train_b64_z4_kernel_stats.md "bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline, session Z4 (round 3, HEAD): fused elementwise kernels + own clip/Adam + dW split heuristic" | head -12 | cut -c1-160
