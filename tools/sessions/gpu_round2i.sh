#!/bin/bash
# Round-2 GPU session I: reference-default frame count (Ft = 480) bench + kernel stats
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_ft480.log 2>&1; echo "bench ft480 rc=$?"; tail -1 $O/bench_ft480.log | cut -c1-1200
cd /tmp; rm -rf /tmp/prof480
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof480 -o b -- python $R/bench.py --t-attn 480 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_ft480.log 2>&1; echo "rocprof rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof480 $O/bench_b256_ft480_kernel_stats.md "bench.py --t-attn 480 --steps 2 --warmup 1 (B=256)" | head -30 | cut -c1-170
