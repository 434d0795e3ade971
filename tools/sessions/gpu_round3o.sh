#!/bin/bash
# Round-3 session O: LDS counters of the pipelined GEMM (plain vs K-strided dX / dW) - what holds dW at 0.82 MFMA-busy
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_INSTS_LDS\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_LDS\|SQ_LDS[A-Z_0-9]*" | sort -u | tr '\n' ' ' | tee $O/lds_counters_3o.log; echo
specs=""
for s in fc7 dx dw attn_core; do
  rm -rf /tmp/pmc_lds_$s
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc_lds_$s -o p -- python $R/tools/mfma_pmc.py $s > $O/pmc_lds_$s.log 2>&1; echo "pmc lds $s rc=$?"; tail -2 $O/pmc_lds_$s.log | cut -c1-200
  specs="$specs $s=/tmp/pmc_lds_$s:gemm_pipe"
done
python $R/tools/pmc_summary.py $O/gemm_lds_pmc_o.json $specs > /dev/null
python - <<PY
import json
j = json.load(open('$O/gemm_lds_pmc_o.json'))
for k, v in j.items():
    if v is None: print(k, None); continue
    w = v.get('SQ_WAVE_CYCLES', 1)
    print(k, {a: round(v[a] / w, 4) for a in v if a.startswith('SQ_') and a != 'SQ_WAVE_CYCLES' and not a.endswith('_per_wave_cycle')}, 'us', round(v['avg_duration_us'], 1))
PY
