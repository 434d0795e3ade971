#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== default"; timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep -v "^TopDown\|amdgpu.ids" | cut -c1-110 | tee $O/r05g_default.txt
echo "== GVD_INGEST_NUMA=0"; GVD_INGEST_NUMA=0 timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep -v "^TopDown\|amdgpu.ids" | cut -c1-110 | tee $O/r05g_numa0.txt
echo "== 8 workers"; GVD_TL_WORKERS=8 timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep -v "^TopDown\|amdgpu.ids" | cut -c1-110 | tee $O/r05g_w8.txt
echo "== GVD_COMPACT=0 (dense preamble)"; GVD_COMPACT=0 timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep "captions/s\|allocator"
