#!/bin/bash
# (ran on an intermediate working tree with a GVD_MAPS_ABL compile-time switch in enc_attn_bwd.hip, removed in commit ee0f7ab; output: profiles/r06/maps_ablate_f.txt)
# Round-6 session F: ablations of the maps kernel, form 1 (prebuilt under tools/_bin/): abl1 = no stores, abl2 = neither epilogue arithmetic nor stores
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
{
for f in 1 2; do
echo "== form $f full"; GVD_MAPS_FORM=$f timeout 300 python tools/attn_train_micro.py 64 0 2>&1 | grep -E "bwd maps"
echo "== form $f no stores"; GVD_MAPS_FORM=$f timeout 300 python tools/with_cflags.py abl1 "-DGVD_MAPS_ABL=1" tools/attn_train_micro.py 64 0 2>&1 | grep -E "bwd maps"
echo "== form $f products only"; GVD_MAPS_FORM=$f timeout 300 python tools/with_cflags.py abl2 "-DGVD_MAPS_ABL=2" tools/attn_train_micro.py 64 0 2>&1 | grep -E "bwd maps"
done
} 2>&1 | tee $O/r06f_maps_ablate.txt
