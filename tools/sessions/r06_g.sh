set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r06g_gpu_tests.txt 2>&1; echo "full suite rc=$?"; tail -3 $O/r06g_gpu_tests.txt | cut -c1-300; grep -E "^FAILED|^ERROR" $O/r06g_gpu_tests.txt | head -20 | cut -c1-250
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06g_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r06g_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print(j['value'], j['ms_per_step'], 'lib', j.get('library_gemms'), 'roofline', r['frac'], r['frac_algorithmic'], r['avg_launch_us'], '| mfma', j['roofline_mfma']['frac'], '| parity', j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'])
c = j['config']
print('b4', c['configs1_b4'])
t = c['configs2_train_b64']; print('train', t['segments_per_s'], t['ms_per_step'], t['parity']['within_1e-4'], t['roofline']['frac'])
b = c['configs4_beam5_t20_b64']; print('beam', b['captions_per_s'], b['ms_per_step'], b['parity']['token_ids_equal'], b['roofline']['frac'], b['roofline']['avg_launch_us'])
f = c['ft480_b256']; print('ft480', f['captions_per_s'], f['ms_per_step'], f['parity']['token_ids_equal'])
"
