#!/bin/bash
# Round-2 GPU session B: masked-proposal compaction - kernel tests, e2e goldens, bench A/B, rocprof.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -m gpu -k "compact or device_side or ragged or padded or pipe or fused_encoder" > $O/test_kernels.log 2>&1; echo "kernels rc=$?"; tail -25 $O/test_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider -m gpu > $O/test_e2e.log 2>&1; echo "e2e rc=$?"; tail -15 $O/test_e2e.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log
env GVD_COMPACT=0 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_dense.log 2>&1; echo "bench dense rc=$?"; tail -1 $O/bench_dense.log | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "rocprof stats rc=$?"
python $R/tools/parse_rocprof.py stats $O/prof_bench $O/bench_kernel_stats.md "bench.py --steps 3 --warmup 1 (B=256 greedy sample)" | head -34
find $O -name "*.db" -delete; find $O -name "*_trace.csv" -size +20M -delete; du -sh $O
