#!/bin/bash
# Round-6 session C: the whole GPU suite on the refactored training path + strict mode, the default bench line (beam section:
# one-exponential group kernel), rocprofv3 kernel statistics of the train step
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -s > $O/r06c_gpu_tests.txt 2>&1; echo "full suite rc=$?"
grep -E "passed|failed" $O/r06c_gpu_tests.txt | tail -2 | cut -c1-300; grep -E "^FAILED|^ERROR" $O/r06c_gpu_tests.txt | head -30 | cut -c1-250
grep -E "torch-native device time|optimiser state after|\|update\||from the fp32 reference|step [0-9]: \|loss" $O/r06c_gpu_tests.txt | cut -c1-400
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06c_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r06c_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print(j['value'], j['ms_per_step'], 'lib', j['library_gemms'], 'roofline', r['frac'], r['frac_algorithmic'], r['frac_physical'], r['avg_launch_us'], '| mfma', j['roofline_mfma']['frac'], '| parity', j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'])
for s in j['roofline_mfma']['per_shape'][:20]: print('   ', s)
c = j['config']
print('b4', c['configs1_b4'])
t = c['configs2_train_b64']; print('train', t['segments_per_s'], t['ms_per_step'], t['parity']['within_1e-4'], t['roofline']['frac'], 'grounding', {k: (t['grounding_stream'] or {}).get(k) for k in ('frac', 'frac_physical', 'avg_launch_us')}, 'compact', t['compacted_rows'].get('compacted_rows'))
b = c['configs4_beam5_t20_b64']; print('beam', b['captions_per_s'], b['ms_per_step'], b['steps_timed'], b['parity'], b['roofline']['frac'], b['roofline']['frac_algorithmic'], b['roofline']['frac_physical'], b['roofline']['avg_launch_us'])
f = c['ft480_b256']; print('ft480', f['captions_per_s'], f['ms_per_step'], f['parity']['token_ids_equal'])
x = c['files_to_captions_ft480']; print('dp', c.get('configs3_dp_train'))
print('files', {k: x.get(k) for k in ('captions_per_s', 'ingest_alone_segments_per_s', 'decode_alone_captions_per_s', 'fraction_of_the_slower_stage', 'error', 'skipped')})
" || tail -5 $O/r06c_bench.log
cd /tmp
export GVD_STATS_ROWS=60
rm -rf /tmp/prof_train
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/r06c_prof_train.log 2>&1; echo "rocprof train rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_train $O/r06c_train_b64_kernel_stats.md "bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline, session C (round 6)" | sed -n 5,40p | cut -c1-170
