#!/bin/bash
# Round-5 session X: the whole GPU suite, smoke and the driver's bench command at the final HEAD (library build unchanged since
# session T: its PMC / kernel-statistics evidence stays valid)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r05x_gpu_tests.txt 2>&1; echo "full suite rc=$?"; tail -3 $O/r05x_gpu_tests.txt | cut -c1-300
timeout 200 python __graft_entry__.py smoke > $O/r05x_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r05x_smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05x_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r05x_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print(j['value'], j['ms_per_step'], 'roofline', r['frac'], r['frac_algorithmic'], r['frac_physical'], r['traffic'], r['avg_launch_us'], '| mfma', j['roofline_mfma']['frac'], '| parity', j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'])
c = j['config']
print('b4', c['configs1_b4'])
t = c['configs2_train_b64']; g = t['grounding_stream']; print('train', t['segments_per_s'], t['ms_per_step'], t['parity']['within_1e-4'], t['roofline']['frac'], 'grounding', [(k, x.get('frac'), x.get('frac_physical'), x.get('avg_launch_us')) for k, x in (('fwd', g), ('d_words', g['backward_d_words']), ('d_regions', g['backward_d_regions']))], 'compact', t['compacted_rows'].get('compacted_rows'))
b = c['configs4_beam5_t20_b64']; print('beam', b['captions_per_s'], b['ms_per_step'], b['steps_timed'], b['parity']['token_ids_equal'], b['roofline']['frac'], b['roofline']['frac_algorithmic'], b['roofline']['frac_physical'])
f = c['ft480_b256']; print('ft480', f['captions_per_s'], f['ms_per_step'], f['parity']['token_ids_equal'])
x = c['files_to_captions_ft480']; print('dp', {k: c['configs3_dp_train'].get(k) for k in ('segments_per_s', 'ms_per_step')})
print('files', {k: x.get(k) for k in ('captions_per_s', 'ingest_alone_segments_per_s', 'decode_alone_captions_per_s', 'fraction_of_the_slower_stage', 'reader_threads', 'cpu_budget', 'read', 'error', 'skipped')})
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
