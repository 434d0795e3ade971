#!/bin/bash
# Round-4 session B: the new parity tests (post-training weights vs the oracle, pipelined eval_split, files -> captions section,
# greedy T=20 golden), dropout-deviation study of the compacted training layout, Newton-step A/B of the score tanh (greedy and
# beam attention kernels), ablation timings of the backward maps kernel, kernel statistics of the train step at HEAD
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ingest.py -q -p no:cacheprovider --timeout=400 -s -k "real_optimisation or eval_split or files_to_captions or t20 or tanh" > $O/r04b_tests.txt 2>&1; echo "tests rc=$?"; grep -E "region logits|passed|failed|Error|error" $O/r04b_tests.txt | cut -c1-400 | tail -12
timeout 200 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -s -k "tanh_fast or attention_step or beam_group" > $O/r04b_tanh.txt 2>&1; echo "tanh tests rc=$?"; grep -E "tanh|passed|failed" $O/r04b_tanh.txt | tail -5
timeout 400 python tools/compact_dropout_study.py 96 8 > $O/r04b_compact_dropout_study.json 2> $O/r04b_compact_dropout_study.err; echo "dropout study rc=$?"; python - <<PY
import json
try:
    j = json.load(open('$O/r04b_compact_dropout_study.json'))
    print(j['summary']); print(j['losses']); print([(x['param'], round(x['ratio'], 2), '%.2e' % x['bias']) for x in j['largest_bias_over_noise']])
except Exception as e:
    print('study output unreadable', e); print(open('$O/r04b_compact_dropout_study.err').read()[-800:])
PY
for v in 1 0; do
  for shape in "256 10 10" "64 10 10 2000 5"; do
    timeout 300 python tools/with_cflags.py newton$v "-DGVD_TANH_NEWTON=$v" tools/profile_attn.py $shape 2>&1 | grep -E "attention_step|with_cflags" | tee -a $O/r04b_newton_ab.log
  done
done
for a in 0 1 2 3 4 7; do
  timeout 300 python tools/with_cflags.py bwdabl$a "-DGVD_BWD_ABL=$a" tools/bwd_maps_micro.py 2>&1 | grep -E "bwd maps|with_cflags" | tee -a $O/r04b_bwd_ablate.log
done
cd /tmp; export GVD_STATS_ROWS=60; rm -rf /tmp/prof_train
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/r04b_prof_train.log 2>&1; echo "rocprof rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_train $O/r04b_train_b64_kernel_stats.md "bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline, session B (round 4): flash-style training attention core" | head -24 | cut -c1-170
tail -1 $O/r04b_prof_train.log | cut -c1-300
