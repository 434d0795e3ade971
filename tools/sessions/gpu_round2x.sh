#!/bin/bash
# Round-2 GPU session X: reference-default frame count (Ft=480) - bench + per-dispatch groups
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_ft480.log 2>&1; echo "bench ft480 rc=$?"; tail -1 $O/bench_ft480.log | cut -c1-330
cd /tmp; rm -rf /tmp/prof_480
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_480 -o f -- python $R/bench.py --t-attn 480 --steps 2 --warmup 1 --no-cpu-baseline --batch 256 > $O/prof_480.log 2>&1; echo "rocprof rc=$?"
python $R/tools/parse_rocprof.py trace /tmp/prof_480 $O/bench_b256_ft480_x_trace.md "bench.py --t-attn 480 --steps 2 --warmup 1 (B=256), session X" | head -32 | cut -c1-180
