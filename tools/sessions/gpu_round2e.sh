#!/bin/bash
# Round-2 GPU session E: attention kernel without the masked rows - tests, micro, PMC traffic, bench, rocprof
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_train.py -q -p no:cacheprovider -m gpu > $O/test.log 2>&1; echo "tests rc=$?"; tail -6 $O/test.log
timeout 300 python tools/profile_attn.py 256 10 10 > $O/attn_micro.log 2>&1; tail -2 $O/attn_micro.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-1500
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/tools/profile_attn.py 256 10 3 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/tools/profile_attn.py 256 10 3 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
python $R/tools/parse_rocprof.py pmc $O/pmc_fetch attn_partial $O/attn_pmc_fetch.json FETCH_SIZE
python $R/tools/parse_rocprof.py pmc $O/pmc_write attn_partial $O/attn_pmc_write.json WRITE_SIZE
python $R/tools/make_attn_traffic.py $O/attn_pmc_fetch.json $O/attn_pmc_write.json $O/attn_traffic.json | tail -8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "rocprof stats rc=$?"
python $R/tools/parse_rocprof.py stats $O/prof_bench $O/bench_kernel_stats.md "bench.py --steps 3 --warmup 1 (B=256 greedy sample)" | sed -n 5,14p | cut -c1-160
find $O -name "*.db" -delete; find $O -name "*_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete
