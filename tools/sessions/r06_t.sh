#!/bin/bash
# Round-6 session T (evidence at the FINAL HEAD; no kernel source changes after this): the stream-kernel tests first (abort if
# they fail), PMC passes of both attention kernels -> profiles/attn_traffic.json keyed on
# this library build; rocprofv3 kernel statistics of the default / beam / Ft=480 / train commands; the whole GPU suite; smoke;
# the driver's bench command
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_stream_mm.py -q -p no:cacheprovider --timeout=200 > $O/r06t_stream_tests.txt 2>&1 || { tail -20 $O/r06t_stream_tests.txt; echo "stream tests failed: abort"; exit 1; }
tail -1 $O/r06t_stream_tests.txt
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"
pmc() {  # tag counters... -- command
  tag=$1; shift; ctr=""; while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > $O/r06t_pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
}
pmc attn_sq $SQ -- python $R/tools/profile_attn.py 256 10 3
pmc attn_fetch FETCH_SIZE -- python $R/tools/profile_attn.py 256 10 3
pmc attn_write WRITE_SIZE -- python $R/tools/profile_attn.py 256 10 3
pmc beam_sq $SQ -- python $R/tools/profile_attn.py 64 10 3 2000 5
pmc beam_fetch FETCH_SIZE -- python $R/tools/profile_attn.py 64 10 3 2000 5
pmc beam_write WRITE_SIZE -- python $R/tools/profile_attn.py 64 10 3 2000 5
pmc gfwd_fetch FETCH_SIZE -- python $R/tools/stream_mm_bench.py 64 5 grounder_fwd
pmc gfwd_write WRITE_SIZE -- python $R/tools/stream_mm_bench.py 64 5 grounder_fwd
pmc gdw_fetch FETCH_SIZE -- python $R/tools/stream_mm_bench.py 64 5 transposed
pmc gdw_write WRITE_SIZE -- python $R/tools/stream_mm_bench.py 64 5 transposed
pmc gdr_fetch FETCH_SIZE -- python $R/tools/stream_mm_bench.py 64 5 N=2048
pmc gdr_write WRITE_SIZE -- python $R/tools/stream_mm_bench.py 64 5 N=2048
python $R/tools/pmc_summary.py $O/r06t_attn_pmc.json gdw_fetch=/tmp/pmc_gdw_fetch:rows_contract_kernel gdw_write=/tmp/pmc_gdw_write:rows_contract_kernel gdr_fetch=/tmp/pmc_gdr_fetch:rank_update_kernel gdr_write=/tmp/pmc_gdr_write:rank_update_kernel gfwd_fetch=/tmp/pmc_gfwd_fetch:grounder_fwd_kernel gfwd_write=/tmp/pmc_gfwd_write:grounder_fwd_kernel greedy_sq=/tmp/pmc_attn_sq:attn_partial greedy_fetch=/tmp/pmc_attn_fetch:attn_partial greedy_write=/tmp/pmc_attn_write:attn_partial beam_sq=/tmp/pmc_beam_sq:attn_partial_group beam_fetch=/tmp/pmc_beam_fetch:attn_partial_group beam_write=/tmp/pmc_beam_write:attn_partial_group > /dev/null
python $R/tools/make_attn_traffic.py $O/r06t_attn_pmc.json $O/attn_traffic.json session-T > /dev/null && cp $O/attn_traffic.json $R/profiles/attn_traffic.json
python - <<PY
import json
j = json.load(open('$O/attn_traffic.json'))
for k in ('greedy', 'beam', 'grounder_fwd', 'grounder_d_words', 'grounder_d_regions'):
    print(k, {a: j[k][a] for a in ('hbm_bytes_per_launch', 'traffic_over_algorithmic', 'avg_duration_us_under_pmc')})
print('srchash', j['lib_srchash'][:12])
PY
export GVD_STATS_ROWS=45
prof() {  # tag title -- bench args
  tag=$1; title=$2; shift; shift; shift
  rm -rf /tmp/prof_$tag
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py "$@" > $O/r06t_prof_$tag.log 2>&1; echo "rocprof $tag rc=$?"
  python $R/tools/parse_rocprof.py stats /tmp/prof_$tag $O/r06t_${tag}_kernel_stats.md "$title" | sed -n 5,11p | cut -c1-150
}
prof b256 "bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sections, session T (round 6, final HEAD)" -- --steps 10 --warmup 3 --no-cpu-baseline --no-sections
prof beam5_t20_b64 "bench.py --beam 5 --frames 20 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline, session T (round 6, final HEAD)" -- --beam 5 --frames 20 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline
prof ft480_b256 "bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections, session T (round 6, final HEAD)" -- --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections
prof train_b64 "bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline, session T (round 6, final HEAD)" -- --mode train --steps 4 --warmup 2 --no-cpu-baseline
cd $R
timeout 300 python tools/native_op_profile.py train > $O/r06t_native_ops_train.txt 2>&1; echo "native ops rc=$?"; grep -E "torch-native device time" $O/r06t_native_ops_train.txt
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r06t_gpu_tests.txt 2>&1; echo "full suite rc=$?"; tail -3 $O/r06t_gpu_tests.txt | cut -c1-300
timeout 200 python __graft_entry__.py smoke > $O/r06t_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r06t_smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06t_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r06t_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print(j['value'], j['ms_per_step'], 'lib', j.get('library_gemms'), 'roofline', r['frac'], r['frac_algorithmic'], r['frac_physical'], r['traffic'], r['avg_launch_us'], '| mfma', j['roofline_mfma']['frac'], '| parity', j['parity']['token_ids_equal'], j['parity']['attended_region_indices_equal'])
c = j['config']
print('b4', c['configs1_b4'])
t = c['configs2_train_b64']; print('train', t['segments_per_s'], t['ms_per_step'], t['parity']['within_1e-4'], t['roofline']['frac'], 'grounding', {k: (t['grounding_stream'] or {}).get(k) for k in ('frac', 'frac_physical', 'avg_launch_us')}, 'compact', t['compacted_rows'].get('compacted_rows'), 'cpu', (t.get('cpu_baseline') or {}).get('value'))
b = c['configs4_beam5_t20_b64']; print('beam', b['captions_per_s'], b['ms_per_step'], b['steps_timed'], b['parity'], b['roofline']['frac'], b['roofline']['frac_algorithmic'], b['roofline']['frac_physical'], 'cpu', (b.get('cpu_baseline') or {}).get('value'))
f = c['ft480_b256']; print('ft480', f['captions_per_s'], f['ms_per_step'], f['parity']['token_ids_equal'], 'cpu', (f.get('cpu_baseline') or {}).get('value'))
x = c['files_to_captions_ft480']; print('dp', c.get('configs3_dp_train'))
print('files', {k: x.get(k) for k in ('captions_per_s', 'ingest_alone_segments_per_s', 'decode_alone_captions_per_s', 'fraction_of_the_slower_stage', 'error', 'skipped')})
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
