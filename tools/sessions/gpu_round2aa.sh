#!/bin/bash
# Round-2 GPU session AA: configs[3] per-GPU shape (32 segments/GPU) and the cost of the gradient-reducer path on one rank
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --mode train --batch 32 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_train_b32.log 2>&1; echo "train B=32 rc=$?"; tail -1 $O/bench_train_b32.log | cut -c1-260
GVD_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --mode train --batch 32 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_train_b32_dp.log 2>&1; echo "train B=32 with the reducer (1 rank, RCCL all-reduce on itself) rc=$?"; tail -1 $O/bench_train_b32_dp.log | cut -c1-260
