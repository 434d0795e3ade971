#!/bin/bash
# Round-3 session G: beam attention kernel with the queries in LDS / 6 waves per SIMD / deeper loads; greedy kernel back at 8 waves
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "beam or greedy_matches" 2>&1 | tail -3
for c in 50 64; do GVD_ATTN_CHUNK=$c timeout 120 python tools/profile_attn.py 256 10 20 2>&1 | tail -1; GVD_ATTN_CHUNK=$c timeout 120 python tools/profile_attn.py 64 10 20 2000 5 2>&1 | tail -1; done | tee $O/attn_ab_3g.log
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam_3g.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam_3g.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'], j['roofline']['frac'])"
timeout 600 python bench.py --no-cpu-baseline --no-sections > $O/bench_3g.log 2>&1; tail -1 $O/bench_3g.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'], j['roofline']['frac'], j['parity'])"
