#!/bin/bash
# quick session: IoU debug + rocprof csv only
set -u
R=$PWD
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/debug_iou.py 2>&1 | tail -12
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof stats rc=$?"
python $R/tools/parse_rocprof.py stats $R/gpurun_out/prof_bench $R/gpurun_out/bench_kernel_stats.md "bench.py --steps 3 --warmup 1 (B=256 greedy sample)" | head -60
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python $R/tools/profile_attn.py 256 10 3 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o p -- python $R/tools/profile_attn.py 256 10 3 > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python $R/tools/parse_rocprof.py pmc $R/gpurun_out/pmc_fetch attn_partial $R/gpurun_out/attn_pmc_fetch.json FETCH_SIZE
python $R/tools/parse_rocprof.py pmc $R/gpurun_out/pmc_write attn_partial $R/gpurun_out/attn_pmc_write.json WRITE_SIZE
find $R/gpurun_out/pmc_fetch $R/gpurun_out/prof_bench -type f | head -20
head -3 $(find $R/gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1)
find $R/gpurun_out -name "*.db" -delete; find $R/gpurun_out -name "*_trace.csv" -size +20M -delete
