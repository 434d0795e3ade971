#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
GVD_TL_EXPERIMENT=1 timeout 600 python tools/files_timeline.py 512 64 3 2 > $O/r05f_timeline_512.txt 2>&1; cat $O/r05f_timeline_512.txt | grep -v "^TopDown\|amdgpu.ids"
nproc; numactl -H 2>/dev/null | head -8; cat /sys/devices/system/node/node*/cpulist 2>/dev/null | head
