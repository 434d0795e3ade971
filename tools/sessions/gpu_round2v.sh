#!/bin/bash
# Round-2 GPU session V: benches after the flash softmax / skew work (decode B=256, beam=5 x 20 frames) + MFMA-busy PMC of the flash kernel
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "flash or compact or encoder" > $O/test_flash.log 2>&1; echo "flash tests rc=$?"; tail -2 $O/test_flash.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_v.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_v.log | cut -c1-330
timeout 600 python bench.py --beam 5 --frames 20 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_beam_v.log 2>&1; echo "bench beam rc=$?"; tail -1 $O/bench_beam_v.log | cut -c1-300
timeout 300 python tools/flash_ablate.py 2>&1 | grep "^ablate" > $O/flash_ablate_v.log; cat $O/flash_ablate_v.log
cd /tmp; rm -rf /tmp/pmc_flash
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_flash -o f -- env FA_CHILD=1 python $R/tools/flash_ablate.py > $O/pmc_flash.log 2>&1; echo "pmc rc=$?"
python $R/tools/parse_rocprof.py pmc /tmp/pmc_flash flash_attn_pad $O/flash_mfma_pmc_v.json SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
