#!/bin/bash
# Round-2 GPU session R: smoke + full GPU suite + default bench (+ under torch.distributed.run with one rank) on HEAD
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_tests_r.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/gpu_tests_r.txt
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_default.log | cut -c1-1500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun1.log 2>&1; echo "bench under torchrun rc=$?"; tail -1 $O/bench_torchrun1.log | cut -c1-300
