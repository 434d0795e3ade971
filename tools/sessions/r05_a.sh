#!/bin/bash
# round 5, session A: the new kernels (stream_mm / gemm_dxs / train_rows) - correctness, then their micro-benchmark, then the
# training tests that run through them, then the train / beam / headline bench lines
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream_mm.py -x -q 2>&1 | tail -25 > $O/r05a_stream_tests.txt; cat $O/r05a_stream_tests.txt
timeout 300 python tools/stream_mm_bench.py 64 20 > $O/r05a_stream_bench.jsonl 2> $O/r05a_stream_bench.err; cat $O/r05a_stream_bench.jsonl; tail -3 $O/r05a_stream_bench.err
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_dist.py -x -q 2>&1 | tail -15 > $O/r05a_train_tests.txt; cat $O/r05a_train_tests.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "beam_group or grounder" 2>&1 | tail -5 > $O/r05a_kernel_tests.txt; cat $O/r05a_kernel_tests.txt
timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $O/r05a_bench_train.log 2>&1; tail -1 $O/r05a_bench_train.log | cut -c1-600
