#!/bin/bash
# (ran on an intermediate working tree: the three forms were selected by a GVD_MAPS_FORM switch and checked by tools/maps_form_check.py, both removed once form 1 was adopted - commit ee0f7ab; output: profiles/r06/maps_forms_e.txt)
# Round-6 session E: the three forms of the backward maps kernel of the training encoder (GVD_MAPS_FORM = 0: round 5's kernel,
# 1: row keys hashed once into LDS + branch-free epilogue + stores straight from the accumulator layout, 2: the same, one
# workgroup walking the key tiles of its query tile) - bitwise agreement of the two maps, the attention-core tests, timings
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
{
for f in 0 1 2; do GVD_MAPS_FORM=$f timeout 300 python tools/maps_form_check.py 2>&1 | grep -E "form=|Error|error|assert" ; done
for f in 0 1 2 0 1 2; do echo "== GVD_MAPS_FORM=$f"; GVD_MAPS_FORM=$f timeout 300 python tools/attn_train_micro.py 64 0 2>&1 | grep -E "maps|core forward"; done
for f in 1 2; do echo "== tests GVD_MAPS_FORM=$f"; GVD_MAPS_FORM=$f timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_fused.py -m gpu -q -p no:cacheprovider -k "enc_attn or encoder" 2>&1 | tail -3; done
} 2>&1 | tee $O/r06e_maps_forms.txt
