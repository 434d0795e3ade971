#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
timeout 100 python tools/stream_mm_bench.py 64 20 grounder_fwd 2>/dev/null | cut -c1-120 | sed 's/^/WIDE4X /'
GVD_GS_WIDE3=1 timeout 100 python tools/stream_mm_bench.py 64 20 grounder_fwd 2>/dev/null | cut -c1-120 | sed 's/^/WIDE3  /'
GVD_GS_NARROW=1 timeout 100 python tools/stream_mm_bench.py 64 20 grounder_fwd 2>/dev/null | cut -c1-120 | sed 's/^/NARROW /'
done | tee $O/r05l_variants.txt
GVD_GS_WIDE3=1 timeout 300 python -m pytest tests/test_gpu_stream_mm.py -x -q -k grounder 2>&1 | tail -1
