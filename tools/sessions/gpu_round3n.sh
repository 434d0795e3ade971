#!/bin/bash
# Round-3 session N: the driver's own invocations of bench.py (plain and under torch.distributed.run with one rank)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_3n.log 2>&1 ) 2>&1 | grep real; echo "bench rc=$?"; tail -1 $O/bench_driver_3n.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_physical'), j['roofline_mfma']['frac'], j['parity']['token_ids_equal'])
c = j['config']
print('b4', c['configs1_b4']); print('train', {k: v for k, v in c['configs2_train_b64'].items() if k != 'roofline'}, c['configs2_train_b64']['roofline']['frac'])
print('beam', {k: v for k, v in c['configs4_beam5_t20_b64'].items() if k != 'roofline'}, c['configs4_beam5_t20_b64']['roofline']['frac'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_torchrun1_3n.log 2>&1 ) 2>&1 | grep real; tail -1 $O/bench_torchrun1_3n.log | cut -c1-250
