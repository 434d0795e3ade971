#!/bin/bash
# Round-6 session D: A/B of the beam-group attention kernel variants (prebuilt under tools/_bin/ by tools/with_cflags.py):
# v0 exponent of the sum per beam (round 5's arithmetic), v1 one exponential per element (product), v2..v5 occupancy /
# scheduling / correction ablations, v6 = v0 with the SLP vectoriser on (round 5's exact build flags)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
run() { tag=$1; fl=$2; shift; shift; for i in 1 2; do "$@" python tools/with_cflags.py $tag "$fl" tools/profile_attn.py 64 10 30 2000 5 2>&1 | grep -E "partial kernel alone|attention_step" | tr '\n' ' '; echo " [$tag]"; done; }
{
run v0_old "-DGVD_GROUP_SPLIT=0 -DGVD_GROUP_WAVES=6" env
run v1_new "-DGVD_GROUP_WAVES=5" env
run v2_w4_nosb "-DGVD_GROUP_WAVES=4 -DGVD_GROUP_SB=0" env
run v3_nocorr "-DGVD_GROUP_NOCORR=1" env
run v4_w6 "-DGVD_GROUP_WAVES=6" env
run v5_old_w5 "-DGVD_GROUP_SPLIT=0 -DGVD_GROUP_WAVES=5" env
run v6_old_slp "-DGVD_GROUP_SPLIT=0 -DGVD_GROUP_WAVES=6" env GVD_VARIANT_NOEXTRA=attention.hip
run v1_new "-DGVD_GROUP_WAVES=5" env
run v0_old "-DGVD_GROUP_SPLIT=0 -DGVD_GROUP_WAVES=6" env
} | tee $O/r06d_beam_variants.txt
timeout 900 python -m pytest tests/test_gpu_strict.py "tests/test_gpu_train.py::test_optimisation_trajectory_matches_reference" "tests/test_gpu_train.py::test_mle_edge_shapes_match_oracle" -m gpu -q -p no:cacheprovider --timeout=600 -s > $O/r06d_tests.txt 2>&1; echo "tests rc=$?"
grep -E "passed|failed" $O/r06d_tests.txt | tail -2 | cut -c1-300; grep -E "^FAILED|^ERROR" $O/r06d_tests.txt | head -20 | cut -c1-250
grep -E "torch-native device time|optimiser state after|\|update\||worst elementwise" $O/r06d_tests.txt | cut -c1-400
