#!/bin/bash
# Round-3 session Z1: fused elementwise kernels of the training step + own clip/Adam - unit tests, train bench A/B (fused vs
# the ATen passes + torch's fused Adam), full GPU suite, the driver's bench command (with the new Ft=480 section)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_train_fused.py -q -p no:cacheprovider --timeout=120 > $O/fused_tests_3z1.txt 2>&1; echo "fused pytest rc=$?"; tail -25 $O/fused_tests_3z1.txt
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_fused_3z1.log 2>&1; echo "bench fused rc=$?"; tail -1 $O/bench_train_fused_3z1.log | cut -c1-300
GVD_TRAIN_FUSED_ELEMENTWISE=0 GVD_OWN_ADAM=0 timeout 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_aten_3z1.log 2>&1; echo "bench aten rc=$?"; tail -1 $O/bench_train_aten_3z1.log | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout=600 --deselect tests/test_gpu_train_fused.py > $O/gpu_tests_3z1.txt 2>&1; echo "pytest rc=$?"; tail -30 $O/gpu_tests_3z1.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_3z1.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3z1.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_physical'), j['roofline']['avg_launch_us'], j['roofline_mfma']['frac'], j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'])
c = j['config']
print('b4', c['configs1_b4']); print('train', c['configs2_train_b64']['segments_per_s'], c['configs2_train_b64']['ms_per_step'], c['configs2_train_b64']['parity']['within_1e-4'], c['configs2_train_b64']['roofline']['frac'])
print('beam', c['configs4_beam5_t20_b64']['captions_per_s'], c['configs4_beam5_t20_b64']['ms_per_step'], c['configs4_beam5_t20_b64']['parity'], c['configs4_beam5_t20_b64']['roofline']['frac'])
print('ft480', c['ft480_b256'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
