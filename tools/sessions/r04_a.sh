#!/bin/bash
# Round-4 session A: first device run of the rebuilt training attention core (flash forward + backward maps kernel), the
# advisor fixes, the timeout / contract recompute paths, the reduced knob surface; the whole GPU suite; the training files
# again on the compacted training layout (GVD_TRAIN_COMPACT=1); micro-benchmarks; the driver's bench command
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout=200 -k "enc_attn or enc_dropout or encoder_training or fused_encoder or beam_group or flash_attention" > $O/r04a_newkernels.txt 2>&1; echo "new-kernel tests rc=$?"; tail -15 $O/r04a_newkernels.txt | cut -c1-300
timeout 200 python tools/attn_train_micro.py > $O/r04a_attn_micro.log 2>&1; echo "micro rc=$?"; cat $O/r04a_attn_micro.log | grep -v Warning | tail -14
timeout 300 python tools/flash_glds_ab.py > $O/r04a_flash_glds_ab.log 2>&1; echo "glds ab rc=$?"; grep -v Warning $O/r04a_flash_glds_ab.log | tail -22
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r04a_gpu_tests.txt 2>&1; echo "full suite rc=$?"; tail -25 $O/r04a_gpu_tests.txt | cut -c1-400
GVD_TRAIN_COMPACT=1 timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py tests/test_gpu_train_fused.py -q -p no:cacheprovider --timeout=300 > $O/r04a_train_tests_compact.txt 2>&1; echo "pytest (compact) rc=$?"; tail -8 $O/r04a_train_tests_compact.txt | cut -c1-300
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04a_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r04a_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('traffic_source'), j['roofline']['avg_launch_us'], j['roofline_mfma']['frac'], j.get('parity'))
c = j['config']
print('b4', c['configs1_b4'])
t = c['configs2_train_b64']; print('train', t['segments_per_s'], t['ms_per_step'], t.get('parity'), t['roofline']['frac'], t.get('compacted_rows'), (t.get('cpu_baseline') or {}).get('value'))
b = c['configs4_beam5_t20_b64']; print('beam', b['captions_per_s'], b['ms_per_step'], b.get('parity'), b['roofline']['frac'], (b.get('cpu_baseline') or {}).get('value'))
f = c['ft480_b256']; print('ft480', f['captions_per_s'], f['ms_per_step'], f.get('parity'), (f.get('cpu_baseline') or {}).get('value'))
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
