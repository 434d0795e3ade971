#!/bin/bash
# files -> captions: CPU quota / scheduling view of the reader threads, the two read modes, fewer reader threads
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
{
  echo "nproc $(nproc); lscpu:"; lscpu | grep -i "^CPU(s)\|Thread\|Socket\|NUMA\|Model name" 
  for how in mapped pread; do
    echo "== GVD_INGEST_READ=$how, 32 reader threads"; GVD_INGEST_READ=$how timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep -v "^TopDown\|amdgpu.ids\|^allocator\|^CPU time\|^  [0-9]*/[0-9]*(t[0-9]* c" | cut -c1-400
  done
  for w in 16 8; do
    echo "== GVD_INGEST_READ=mapped, $w reader threads"; GVD_TL_WORKERS=$w timeout 300 python tools/files_timeline.py 512 64 3 2 2>&1 | grep -v "^TopDown\|amdgpu.ids\|^allocator\|^CPU time\|^  [0-9]*/[0-9]*(t[0-9]* c" | cut -c1-400
  done
} 2>&1 | tee $O/r05u_quota.txt
