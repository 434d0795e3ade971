#!/bin/bash
# Round-3 session W: kernel statistics at a mid batch (B=32)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_b32
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -o p -- python $R/bench.py --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-sections > $O/prof_b32_3w.log 2>&1; echo "rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_b32 $O/b32_w_kernel_stats.md "bench.py --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-sections, session W (round 3)" | head -34 | cut -c1-150
