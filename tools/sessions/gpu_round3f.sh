#!/bin/bash
# Round-3 session F: PMC evidence (attention kernels: VALU / wait / HBM bytes; MFMA-busy of every GEMM class north_star names),
# B=4 kernel statistics, ingest worker sweep with the native reader, Ft=480 / PCIe-inclusive / DP-forced bench lines
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "bntrain" 2>&1 | tail -2
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"
pmc() {  # tag counters... -- command
  tag=$1; shift; ctr=""; while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
}
pmc attn_sq $SQ -- python $R/tools/profile_attn.py 256 10 3
pmc attn_fetch FETCH_SIZE -- python $R/tools/profile_attn.py 256 10 3
pmc attn_write WRITE_SIZE -- python $R/tools/profile_attn.py 256 10 3
pmc beam_sq $SQ -- python $R/tools/profile_attn.py 64 10 3 2000 5
pmc beam_fetch FETCH_SIZE -- python $R/tools/profile_attn.py 64 10 3 2000 5
pmc beam_write WRITE_SIZE -- python $R/tools/profile_attn.py 64 10 3 2000 5
python $R/tools/pmc_summary.py $O/attn_pmc_f.json greedy_sq=/tmp/pmc_attn_sq:attn_partial greedy_fetch=/tmp/pmc_attn_fetch:attn_partial greedy_write=/tmp/pmc_attn_write:attn_partial beam_sq=/tmp/pmc_beam_sq:attn_partial_group beam_fetch=/tmp/pmc_beam_fetch:attn_partial_group beam_write=/tmp/pmc_beam_write:attn_partial_group > /dev/null
python - <<PY
import json
j = json.load(open('$O/attn_pmc_f.json'))
for k, v in j.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) and b < 100 else b) for a, b in (v or {}).items()})
PY
specs=""
for s in fc7 qkv dx dw attn_core logit attn_hid lstm flash; do
  pmc mfma_$s SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -- python $R/tools/mfma_pmc.py $s
  case $s in logit|attn_hid|lstm) sub=gemm_small;; flash) sub=flash_attn_pad;; *) sub=gemm_pipe;; esac
  specs="$specs $s=/tmp/pmc_mfma_$s:$sub"
done
python $R/tools/pmc_summary.py $O/mfma_pmc_f.json $specs > /dev/null
python - <<PY
import json
j = json.load(open('$O/mfma_pmc_f.json'))
for k, v in j.items():
    print(k, None if v is None else {a: (round(b, 4) if isinstance(b, float) and b < 1e4 else b) for a, b in v.items() if a in ('mfma_busy_frac', 'avg_duration_us', 'launches')})
PY
rm -rf /tmp/prof_b4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b4 -o p -- python $R/bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline > $O/prof_b4_3f.log 2>&1; echo "rocprof b4 rc=$?"
python $R/tools/parse_rocprof.py stats /tmp/prof_b4 $O/b4_f_kernel_stats.md "bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline, session F (round 3)" | head -30 | cut -c1-150
cd $R
timeout 600 python tools/ingest_bench.py 256 128 8,16,32,64 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/ingest_bench_3f.log
timeout 600 python bench.py --t-attn 480 --steps 5 --warmup 2 --no-cpu-baseline --no-sections > $O/bench_ft480_3f.log 2>&1; echo "bench ft480 rc=$?"; tail -1 $O/bench_ft480_3f.log | cut -c1-260
timeout 600 python bench.py --h2d --steps 6 --warmup 2 --no-cpu-baseline --no-sections > $O/bench_h2d_3f.log 2>&1; echo "bench h2d rc=$?"; tail -1 $O/bench_h2d_3f.log | cut -c1-260
GVD_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --mode train --batch 32 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_train_b32_dpforce_3f.log 2>&1; echo "bench dpforce rc=$?"; tail -1 $O/bench_train_b32_dpforce_3f.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], j['dp_bucket_launches'])"
