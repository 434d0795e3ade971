#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for f in grounder_fwd "rows_contract d xt" "rank_update d feats"; do
  timeout 100 python tools/stream_mm_bench.py 64 20 "$f" 2>/dev/null | cut -c1-170
  GVD_STREAM_NOMATH=1 timeout 100 python tools/stream_mm_bench.py 64 20 "$f" 2>/dev/null | cut -c1-170 | sed 's/^/NOMATH /'
done | tee $O/r05j_nomath.txt
