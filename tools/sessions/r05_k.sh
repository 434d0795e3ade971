#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stream_mm.py -x -q 2>&1 | tail -3
for f in grounder_fwd "rows_contract d xt"; do
  timeout 100 python tools/stream_mm_bench.py 64 20 "$f" 2>/dev/null | cut -c1-170
  GVD_STREAM_NOMATH=1 timeout 100 python tools/stream_mm_bench.py 64 20 "$f" 2>/dev/null | cut -c1-170 | sed 's/^/NOMATH /'
done | tee $O/r05k_wide.txt
GVD_GS_NARROW=1 timeout 100 python tools/stream_mm_bench.py 64 20 grounder_fwd 2>/dev/null | cut -c1-170 | sed 's/^/NARROW /' | tee -a $O/r05k_wide.txt
