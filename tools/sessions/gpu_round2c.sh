#!/bin/bash
# Round-2 GPU session C: evidence refresh - new tests, MFMA-busy PMC of the GEMM / attention kernels, train and beam
# bench lines + kernel statistics, HBM traffic of the backward attention kernels.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_train.py -q -p no:cacheprovider -m gpu > $O/test_train_dist.log 2>&1; echo "train+dist tests rc=$?"; tail -12 $O/test_train_dist.log
timeout 600 python bench.py --mode train --steps 5 --warmup 2 > $O/bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/bench_train.log | cut -c1-900
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 > $O/bench_beam.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam.log | cut -c1-900
cd /tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $R/tools/mfma_pmc.py > $O/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
  python $R/tools/parse_pmc_multi.py $O/pmc_mfma $O/mfma_pmc.json
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_train.log 2>&1; echo "rocprof train rc=$?"
python $R/tools/parse_rocprof.py stats $O/prof_train $O/train_b64_kernel_stats.md "bench.py --mode train --steps 2 --warmup 1 (B=64)" | head -30 | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_beam -o beam -- python $R/bench.py --beam 5 --frames 20 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_beam.log 2>&1; echo "rocprof beam rc=$?"
python $R/tools/parse_rocprof.py stats $O/prof_beam $O/beam5_t20_b64_kernel_stats.md "bench.py --beam 5 --frames 20 --batch 64 --steps 2 --warmup 1" | head -16 | cut -c1-160
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_train_$C -o p -- python $R/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_train_$C.log 2>&1; echo "pmc train $C rc=$?"
  python $R/tools/parse_pmc_multi.py $O/pmc_train_$C $O/train_pmc_$C.json attn_bwd_step=attn_bwd_step_kernel attn_bwd_pfeats=attn_bwd_pfeats_kernel attn_partial=attn_partial_kernel
done
find $O -name "*.db" -delete; find $O -name "*_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
