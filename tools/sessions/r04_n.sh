#!/bin/bash
# Round-4 session N: where do the waves of the training attention core's kernels wait?  PMC passes (separate runs, --pmc with
# --kernel-trace only) over tools/attn_train_micro.py 64 0: one-head-slot GEMM (gemm_n192), backward maps kernel, training
# flash forward.  Diagnosis only - no source change rides on this session.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)?\s*:?\s*(SQ_[A-Z0-9_]+|TCC_[A-Z0-9_a-z\[\]]+|TCP_[A-Z0-9_a-z]+|GRBM_[A-Z_]+)" | grep -oE "(SQ|TCC|TCP|GRBM)_[A-Za-z0-9_]+" | sort -u > $O/r04n_counters.txt; wc -l $O/r04n_counters.txt
pmc() {  # tag counters... 
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/attn_train_micro.py 64 0 > $O/r04n_pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
}
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc lds SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA
pmc misc SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
S=""
for t in sq lds mfma misc fetch write tcc; do for k in gemm_n192_kernel enc_attn_bwd_maps flash_attn_pad; do S="$S ${k}__$t=/tmp/pmc_$t:$k"; done; done
python $R/tools/pmc_summary.py $O/r04n_train_core_pmc.json $S > /dev/null; echo "summary rc=$?"
python - <<PY
import json
j = json.load(open('$O/r04n_train_core_pmc.json'))
for k, v in j.items():
    if v: print(k, {a: (round(b, 4) if b < 100 else round(b)) for a, b in v.items()})
PY
