#!/bin/bash
# Round-2 GPU session A: new kernels (pipelined GEMM, padded-head flash attention) - numerics, micro-benchmarks, the
# reference goldens on the new and on the old path, bench A/B, rocprof kernel stats.  Outputs under gpurun_out/.
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
OLD="GVD_GEMM_VARIANT=1 GVD_ENC_FUSED=0 GVD_POOL_EMBED_OWN=0"
(timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"); tail -2 $O/smoke.log
timeout 600 python tools/gemm_micro.py 1 3 > $O/gemm_micro.log 2>&1; echo "gemm_micro rc=$?"; cat $O/gemm_micro.log | tail -20
timeout 300 python tools/flash_micro.py > $O/flash_micro.log 2>&1; echo "flash_micro rc=$?"; tail -4 $O/flash_micro.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -m gpu > $O/test_kernels.log 2>&1; echo "kernels rc=$?"; tail -15 $O/test_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider -m gpu > $O/test_e2e.log 2>&1; echo "e2e rc=$?"; tail -15 $O/test_e2e.log
env $OLD timeout 600 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider -m gpu -k "greedy_matches_reference or round_trip or beam_search_matches_reference" > $O/test_e2e_oldpath.log 2>&1; echo "e2e old path rc=$?"; tail -8 $O/test_e2e_oldpath.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ingest.py -q -p no:cacheprovider -m gpu > $O/test_rest.log 2>&1; echo "rest rc=$?"; tail -5 $O/test_rest.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log
env $OLD timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_oldpath.log 2>&1; echo "bench old rc=$?"; tail -1 $O/bench_oldpath.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "rocprof stats rc=$?"
python $R/tools/parse_rocprof.py stats $O/prof_bench $O/bench_kernel_stats.md "bench.py --steps 3 --warmup 1 (B=256 greedy sample)" | head -40
find $O -name "*.db" -delete; find $O -name "*_trace.csv" -size +20M -delete; du -sh $O
