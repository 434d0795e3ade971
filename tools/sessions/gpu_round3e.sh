#!/bin/bash
# Round-3 session E: full GPU suite on the new tests, default bench line, fresh rocprofv3 kernel statistics of the default /
# train / beam commands, ingest worker sweep
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_tests_3e.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/gpu_tests_3e.txt
timeout 600 python bench.py > $O/bench_3e.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3e.log | cut -c1-300
cd /tmp
for what in "b256:--no-cpu-baseline --no-sections" "train_b64:--mode train --steps 4 --warmup 2 --no-cpu-baseline" "beam5_t20_b64:--beam 5 --frames 20 --batch 64 --steps 4 --warmup 2 --no-cpu-baseline"; do
  tag=${what%%:*}; args=${what#*:}
  rm -rf /tmp/prof_$tag
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py $args > $O/prof_${tag}_3e.log 2>&1; echo "rocprof $tag rc=$?"
  python $R/tools/parse_rocprof.py stats /tmp/prof_$tag $O/${tag}_e_kernel_stats.md "bench.py $args, session E (round 3)" | head -16 | cut -c1-170
done
cd $R
timeout 600 python tools/ingest_bench.py 256 128 8,16,32,48,64 2>&1 | grep -v Warning | tee $O/ingest_bench_3e.log
