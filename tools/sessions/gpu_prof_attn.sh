#!/bin/bash
# kernel-trace statistics of the default bench + the two PMC passes of the attention kernel
set -u
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof stats rc=$?"
python $R/tools/parse_rocprof.py stats $R/gpurun_out/prof_bench $R/gpurun_out/bench_kernel_stats.md "bench.py --steps 3 --warmup 1 (B=256 greedy sample)" | sed -n 5,14p | cut -c1-150
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python $R/tools/profile_attn.py 256 10 3 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o p -- python $R/tools/profile_attn.py 256 10 3 > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python $R/tools/parse_rocprof.py pmc $R/gpurun_out/pmc_fetch attn_partial $R/gpurun_out/attn_pmc_fetch.json FETCH_SIZE
python $R/tools/parse_rocprof.py pmc $R/gpurun_out/pmc_write attn_partial $R/gpurun_out/attn_pmc_write.json WRITE_SIZE
timeout 200 python $R/bench.py --steps 10 --warmup 2 > $R/gpurun_out/bench.log 2>&1; tail -1 $R/gpurun_out/bench.log | cut -c1-200
find $R/gpurun_out -name "*.db" -delete; find $R/gpurun_out -name "*_trace.csv" -size +20M -delete
