#!/bin/bash
# Round-2 GPU session S: sustained GEMM rate (power-limited clock?) vs the burst rate of the micro-benchmark
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(rocm-smi --showclocks --showpower 2>/dev/null | head -30) > $O/smi_idle.log
timeout 300 python tools/gemm_sustained.py 256000 2048 2048 6 > $O/gemm_sustained.log 2>&1 &
PID=$!
sleep 4.5; (rocm-smi --showclocks --showpower 2>/dev/null | head -30) > $O/smi_load.log
wait $PID; cat $O/gemm_sustained.log
timeout 300 python tools/gemm_sustained.py 205000 1024 2784 3 >> $O/gemm_sustained.log 2>&1; tail -2 $O/gemm_sustained.log
grep -i "sclk\|power\|mclk" $O/smi_idle.log | head -8; echo ---; grep -i "sclk\|power\|mclk" $O/smi_load.log | head -8
