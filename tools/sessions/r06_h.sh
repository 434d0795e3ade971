set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; export GVD_STATS_ROWS=60
rm -rf /tmp/prof_b4
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b4 -o p -- python $R/bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-sections > $O/r06h_prof_b4.log 2>&1; echo "rocprof rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_b4 $O/r06h_b4_kernel_stats.md "bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-sections (round 6, session H)" | sed -n 5,50p | cut -c1-160
tail -1 $O/r06h_prof_b4.log | cut -c1-300
