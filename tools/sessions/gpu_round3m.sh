#!/bin/bash
# Round-3 session M: gradient-ready bucket order (DP launch timeline), two more MFMA PMC sections, full GPU suite + smoke + default bench on HEAD
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke > $O/smoke_3m.log 2>&1; echo "smoke rc=$?"); tail -1 $O/smoke_3m.log
GVD_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --mode train --batch 32 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_b32_dpforce_3m.log 2>&1; echo "bench dpforce rc=$?"; tail -1 $O/bench_train_b32_dpforce_3m.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], j['dp_bucket_launches'])"
cd /tmp
specs=""
for s in ctx2pool logit_train; do
  rm -rf /tmp/pmc_mfma_$s
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma_$s -o p -- python $R/tools/mfma_pmc.py $s > $O/pmc_mfma_$s.log 2>&1; echo "pmc $s rc=$?"
  specs="$specs $s=/tmp/pmc_mfma_$s:gemm_pipe"
done
python $R/tools/pmc_summary.py $O/mfma_pmc_m.json $specs | grep -i "mfma_busy_frac\|avg_duration\|\"ctx2pool\|\"logit"
cd $R
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gpu_tests_3m.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests_3m.txt
timeout 600 python bench.py > $O/bench_3m.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_3m.log | cut -c1-200
