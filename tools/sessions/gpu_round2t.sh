#!/bin/bash
# Round-2 GPU session T: issue-order timeline of the big kernels of the B=256 step (are the GEMMs slower after light phases?)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_inf
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_inf -o inf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 256 > $O/prof_inf.log 2>&1; echo "rocprof inf rc=$?"
python $R/tools/parse_rocprof.py timeline /tmp/prof_inf $O/bench_b256_t_timeline.md 250 | tail -75 | cut -c1-150
