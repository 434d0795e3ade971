#!/bin/bash
# PMC passes over the flash attention kernel (default 16x16x4 variant): MFMA busy, VALU / LDS / wait breakdown
set -u
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/fpmc_$tag -o p -- python $R/tools/mfma_pmc.py > $R/gpurun_out/fpmc_$tag.log 2>&1; echo "$C rc=$?"
  python $R/tools/parse_pmc_multi.py $R/gpurun_out/fpmc_$tag $R/gpurun_out/fpmc_$tag.json 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin)
print({k: {n: round(v, 1) for n, v in c.items()} for k, c in d.items() if k == 'flash_attn'})"
done
find $R/gpurun_out -name "*.db" -delete; find $R/gpurun_out -name "*_trace.csv" -size +5M -delete
