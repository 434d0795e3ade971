#!/bin/bash
# kernel-trace statistics of the batch-4 (BASELINE configs[1]) greedy path
set -u
R=$PWD
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b4 -o bench -- python $R/bench.py --batch 4 --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_b4.log 2>&1; echo "rocprof stats rc=$?"
tail -1 $R/gpurun_out/prof_b4.log
python $R/tools/parse_rocprof.py stats $R/gpurun_out/prof_b4 $R/gpurun_out/bench_b4_kernel_stats.md "bench.py --batch 4 --steps 20 --warmup 3 (B=4 greedy sample, 23 calls)" | head -50
find $R/gpurun_out -name "*.db" -delete; find $R/gpurun_out -name "*_trace.csv" -size +20M -delete
