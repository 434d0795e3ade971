#!/bin/bash
# Round-2 GPU session O: fused beam-step kernel
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "beam" > $O/test_beam.log 2>&1; echo "beam tests rc=$?"; tail -6 $O/test_beam.log
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam.log | cut -c1-300
GVD_BEAM_FUSED=0 timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam_torch.log 2>&1; echo "bench beam (torch bookkeeping) rc=$?"; tail -1 $O/bench_beam_torch.log | cut -c1-300
cd /tmp; rm -rf /tmp/prof_beam
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_beam -o beam -- python $R/bench.py --beam 5 --frames 20 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_beam.log 2>&1; echo "rocprof beam rc=$?"
python $R/tools/parse_rocprof.py trace /tmp/prof_beam $O/beam5_t20_b64_o_trace.md "bench.py --beam 5 --frames 20 --batch 64 --steps 1 --warmup 1, session O: per-dispatch groups" | head -45 | cut -c1-200
