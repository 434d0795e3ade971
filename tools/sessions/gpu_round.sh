#!/bin/bash
# One GPU-box session: tests, bench, rocprofv3 kernel stats, PMC passes.  Outputs under gpurun_out/.
set -u
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"); tail -2 gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/test.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench.log
timeout 300 python bench.py --steps 20 --warmup 3 --batch 4 --no-cpu-baseline > gpurun_out/bench_b4.log 2>&1; tail -1 gpurun_out/bench_b4.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof stats rc=$?"
python $R/tools/parse_rocprof.py stats $R/gpurun_out/prof_bench $R/gpurun_out/bench_kernel_stats.md "bench.py --steps 3 --warmup 1 (B=256 greedy sample)" | head -50
timeout 300 python $R/tools/profile_attn.py 256 10 10 > $R/gpurun_out/attn_micro.log 2>&1; cat $R/gpurun_out/attn_micro.log | tail -2
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python $R/tools/profile_attn.py 256 10 3 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o p -- python $R/tools/profile_attn.py 256 10 3 > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python $R/tools/parse_rocprof.py pmc $R/gpurun_out/pmc_fetch attn_partial $R/gpurun_out/attn_pmc_fetch.json FETCH_SIZE
python $R/tools/parse_rocprof.py pmc $R/gpurun_out/pmc_write attn_partial $R/gpurun_out/attn_pmc_write.json WRITE_SIZE
find $R/gpurun_out/pmc_fetch $R/gpurun_out/prof_bench -type f | head -20; du -sh $R/gpurun_out
# keep the merged-back payload small
find $R/gpurun_out -name "*.db" -delete; find $R/gpurun_out -name "*_trace.csv" -size +20M -delete
