#!/bin/bash
# Round 5, first call: the workgroup-shared LGA ring (GANET_LGA_WG=1: lga_apply_pp_wx / _wxo, a barrier per plane pair; =2:
# lga_apply_pp_fx / _fxo, progress flags; built and emulator-verified in round 4, never run on a GPU).  (1) its parity tests, under a timeout of their own: a hang here must not take the call with it;
# (2) whole-step A/B against the default kernels on one box (hipGraph replay, interleaved: +-0.2 %), ring depths 5 / 8 / 10 if the
# variant libraries were built (python scripts/build_variants.py wg5:-DLGAP_WG_NR=5 wg10:-DLGAP_WG_NR=10);
# (3) the LGA kernels of the step one by one, same settings.     bash scripts/gpu_r5_wg.sh <tag>
TAG=${1:-r8a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
timeout -k 5 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "workgroup_ring" > $OUT/tests_wg.log 2>&1; echo "wg tests rc=$?"; tail -3 $OUT/tests_wg.log
LIBS="libganet_hip.so libganet_hip.so@GANET_LGA_WG=1 libganet_hip.so@GANET_LGA_WG=2"
for V in wg5 wg10; do [ -f ganet_amd/libganet_hip_$V.so ] && LIBS="$LIBS libganet_hip_$V.so@GANET_LGA_WG=1 libganet_hip_$V.so@GANET_LGA_WG=2"; done
timeout -k 5 300 python scripts/ab_step.py $LIBS > $OUT/ab_step_wg.txt 2>&1; echo "ab_step rc=$?"; tail -8 $OUT/ab_step_wg.txt
timeout -k 5 200 python scripts/ab_lga_stages.py $LIBS > $OUT/ab_lga_stages_wg.txt 2>&1; echo "ab_lga_stages rc=$?"; tail -12 $OUT/ab_lga_stages_wg.txt
