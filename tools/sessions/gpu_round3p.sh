#!/bin/bash
# Round-3 session P: inference frame embeddings without concat / BatchNorm transposes (Ft=480), beam chunk size
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('Ft=480 B=256:', j['value'], j['ms_per_step'])" | tee $O/ft480_3p.log
timeout 600 python bench.py --t-attn 480 --batch 4 --steps 20 --warmup 3 --no-cpu-baseline --no-sections 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('Ft=480 B=4:', j['value'], j['ms_per_step'])" | tee -a $O/ft480_3p.log
for c in 50 64; do
GVD_ATTN_CHUNK=$c timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('beam chunk=$c', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'], j['roofline']['frac'])"
done | tee $O/beam_chunk_3p.log
