#!/bin/bash
# round 5, session B: second pass over the new kernels - split-K fixup with parallel loads, rows_contract through the transposed
# S copy, grounder_fwd ring / tile variants (GVD_GS_VARIANT), then PMC passes (separate runs) on the micro-benchmark
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stream_mm.py -x -q 2>&1 | tail -5 > $O/r05b_stream_tests.txt; cat $O/r05b_stream_tests.txt
timeout 300 python tools/stream_mm_bench.py 64 20 > $O/r05b_stream_bench.jsonl 2> $O/r05b_stream_bench.err; cat $O/r05b_stream_bench.jsonl | cut -c1-260; tail -2 $O/r05b_stream_bench.err
for v in 1 2 3; do GVD_GS_VARIANT=$v timeout 120 python tools/stream_mm_bench.py 64 20 grounder_fwd 2>/dev/null | cut -c1-200 | tee -a $O/r05b_gs_variants.jsonl; done
for v in 1 2 3; do GVD_GS_VARIANT=$v timeout 200 python -m pytest tests/test_gpu_stream_mm.py -x -q -k grounder 2>&1 | tail -1; done
cd /tmp
pmc() {  # tag counters...
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/stream_mm_bench.py 64 5 > $O/r05b_pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
}
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc lds SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
S=""
for t in sq lds mfma fetch write tcc tcp; do for k in grounder_fwd_kernel rows_contract_kernel rank_update_kernel gemm_dxs_kernel; do S="$S ${k}__$t=/tmp/pmc_$t:$k"; done; done
python $R/tools/pmc_summary.py $O/r05b_stream_pmc.json $S > /dev/null; echo "summary rc=$?"
python - <<PY
import json
j = json.load(open('$O/r05b_stream_pmc.json'))
for k, v in sorted(j.items()):
    if v: print(k, {a: (round(b, 4) if b < 100 else round(b)) for a, b in v.items()})
PY
