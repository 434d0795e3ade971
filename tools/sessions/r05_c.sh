#!/bin/bash
# round 5, session C: transfer_mode='none' goldens, ingest on the native batch reader, bench line with the configs[3] section,
# files -> captions composition, torch-native share of the train step after the BPTT went library-free
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream_mm.py tests/test_gpu_ingest.py -x -q 2>&1 | tail -4 > $O/r05c_tests1.txt; cat $O/r05c_tests1.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_train.py -x -q -k "tnone or bench_two_ranks or bench_under" 2>&1 | tail -4 > $O/r05c_tests2.txt; cat $O/r05c_tests2.txt
timeout 300 python tools/stream_mm_bench.py 64 20 > $O/r05c_stream_bench.jsonl 2> $O/r05c_stream_bench.err; cut -c1-200 $O/r05c_stream_bench.jsonl
timeout 600 python bench.py --files 256 > $O/r05c_files.log 2>&1; tail -1 $O/r05c_files.log | cut -c1-700
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05c_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r05c_bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
c=j['config']
print('value',j['value'],'ms',j['ms_per_step'],'parity',j.get('parity'))
print('roofline',{k:j['roofline'][k] for k in ('frac','frac_algorithmic','frac_physical','avg_launch_us','bytes_per_launch')})
print('b4',c.get('configs1_b4'))
t=c.get('configs2_train_b64') or {}
print('train',{k:t.get(k) for k in ('segments_per_s','ms_per_step','parity','grounding_stream')})
b=c.get('configs4_beam5_t20_b64') or {}
print('beam',{k:b.get(k) for k in ('captions_per_s','ms_per_step','steps_timed','parity')}, (b.get('roofline') or {}).get('frac'), (b.get('roofline') or {}).get('frac_algorithmic'))
print('ft480',{k:(c.get('ft480_b256') or {}).get(k) for k in ('captions_per_s','parity')})
print('files',c.get('files_to_captions_ft480'))
print('dp',c.get('configs3_dp_train'))
"
timeout 300 python tools/native_op_profile.py train > $O/r05c_native_ops_train.txt 2>&1; head -30 $O/r05c_native_ops_train.txt
