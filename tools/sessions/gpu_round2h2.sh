#!/bin/bash
# Round-2 GPU session H2: training kernel stats, MFMA attention core on / off
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for mode in 1 0; do
  rm -rf /tmp/prof_train
  GVD_ENC_TRAIN_MFMA=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_train_$mode.log 2>&1; echo "rocprof train mfma=$mode rc=$?"
  python $R/tools/parse_rocprof.py stats /tmp/prof_train $O/train_b64_h_mfma${mode}_kernel_stats.md "bench.py --mode train --steps 2 --warmup 1 (B=64), GVD_ENC_TRAIN_MFMA=$mode" | head -34 | cut -c1-170
  cp $(find /tmp/prof_train -name '*kernel_stats.csv' | head -1) $O/train_b64_h_mfma${mode}_kernel_stats.csv
done
