#!/bin/bash
# Round 5, call 2: r8a decided the quads (lost), the fused filter gradient (lost) and the flag rings (lost to the barrier rings); it
# also showed the barrier rings winning on the four apply launches and LOSING on the two filter-gradient launches, and its parity test
# stopped on a bug of the test harness.  This call: (1) device parity of the barrier rings, (2) whole-step A/B of rings on all six
# launches (GANET_LGA_WG=1) against rings on the apply launches only (=3), at ring depths 5 .. 8, (3) the LGA kernels one by one.
TAG=${1:-r8b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
export GANET_TEST_WG=1
GANET_TEST_WG_FORMS=1 timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bounds.py tests/test_gpu_modules.py -q -m gpu \
  -k "workgroup_ring or lga_chain_on_end_aligned or round5" > $OUT/tests_rings.log 2>&1; echo "ring tests rc=$?"; tail -5 $OUT/tests_rings.log
L=libganet_hip.so
LIBS="$L $L@GANET_LGA_WG=1 $L@GANET_LGA_WG=3"
for V in wg5 wg6 wg7; do [ -f ganet_amd/libganet_hip_$V.so ] && LIBS="$LIBS libganet_hip_$V.so@GANET_LGA_WG=3"; done
timeout -k 5 400 python scripts/ab_step.py $LIBS > $OUT/ab_step.txt 2>&1; echo "ab_step rc=$?"; tail -8 $OUT/ab_step.txt
timeout -k 5 200 python scripts/ab_lga_stages.py $L $L@GANET_LGA_WG=3 > $OUT/ab_lga_stages.txt 2>&1; echo "ab_lga_stages rc=$?"; tail -4 $OUT/ab_lga_stages.txt
