#!/bin/bash
# Round 5, first call: the workgroup-shared LGA rings (GANET_LGA_WG=1: a barrier per plane pair; =2: progress flags; all six LGA
# launches of the step; built and emulator-verified in round 4, never run on a GPU).
# (1) their parity tests, under a timeout of their own: a hang here must not take the call with it;
# (2) whole-step A/B against the default kernels on one box (hipGraph replay, interleaved: +-0.2 %), plus ring depths / slack if the
#     variant libraries were built:  python scripts/build_variants.py wg6:-DLGAP_WG_NR=6 wgf6:-DLGAP_WG_NR_FG=6 wg10:-DLGAP_WG_NR=10,-DLGAP_WG_NR_FG=8 wgs2:-DLGAP_WG_NR=10,-DLGAP_WG_SLACK=2,-DLGAP_WG_NR_FG=8
#     (each checked under the emulator first: python scripts/sim_wg_variants.py <flags>; the filter gradient's ring stays at 8 slots: 3 workgroups of 49 KB per CU)
# (3) the LGA kernels of the step one by one, same settings; (4) fabric traffic of the step with the rings on (the point of them:
#     x over-fetch 1.5 - 1.9 x -> ?), one PMC pass.                                  bash scripts/gpu_r5_wg.sh <tag>
TAG=${1:-r8a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
GANET_TEST_WG=1 timeout -k 5 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "workgroup_ring" > $OUT/tests_wg.log 2>&1; echo "wg tests rc=$?"; tail -3 $OUT/tests_wg.log
# the fused two-pass filter gradient (ganet_lga2_filter_grad; GANET_LGA_FG_FUSED=1 in the Python layer): checked inside the paired chains
GANET_TEST_WG=1 timeout -k 5 300 python -m pytest tests/test_gpu_bounds.py tests/test_gpu_parity.py tests/test_gpu_modules.py -x -q -m gpu -k "lga_chain_on_end_aligned or paired or round5 or pixel_quads" > $OUT/tests_fused.log 2>&1; echo "fused-fg tests rc=$?"; tail -3 $OUT/tests_fused.log
LIBS="libganet_hip.so libganet_hip.so@GANET_SGA_POINT_Q4=1 libganet_hip.so@GANET_LGA_FG_FUSED=1 libganet_hip.so@GANET_LGA_WG=1 libganet_hip.so@GANET_LGA_WG=2 libganet_hip.so@GANET_LGA_WG=1,GANET_LGA_MIX=0 libganet_hip.so@GANET_LGA_WG=2,GANET_LGA_SEGS=2"
[ -f ganet_amd/libganet_hip_q4w5.so ] && LIBS="$LIBS libganet_hip_q4w5.so@GANET_SGA_POINT_Q4=1"      # python scripts/build_variants.py q4w5:-DGA_POINT_Q4_WAVES=5
for V in wg6 wgf6 wg10 wgs2; do [ -f ganet_amd/libganet_hip_$V.so ] && LIBS="$LIBS libganet_hip_$V.so@GANET_LGA_WG=1 libganet_hip_$V.so@GANET_LGA_WG=2"; done
timeout -k 5 400 python scripts/ab_step.py $LIBS > $OUT/ab_step_wg.txt 2>&1; echo "ab_step rc=$?"; tail -14 $OUT/ab_step_wg.txt
timeout -k 5 240 python scripts/ab_lga_stages.py libganet_hip.so libganet_hip.so@GANET_LGA_WG=1 libganet_hip.so@GANET_LGA_WG=2 > $OUT/ab_lga_stages_wg.txt 2>&1; echo "ab_lga_stages rc=$?"; tail -14 $OUT/ab_lga_stages_wg.txt
for WG in 0 1; do
  ( cd /tmp; GANET_LGA_WG=$WG timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/pmc_wg$WG/p1 -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py step 3 > $OUT/pmc_wg$WG.log 2>&1; echo "pmc wg=$WG rc=$?" )
  python scripts/pmc_summary.py $OUT/pmc_wg$WG > $OUT/summary_wg$WG.txt 2>&1
  find $OUT/pmc_wg$WG -name '*.csv' -size +1M -delete
done
grep -h "^== lga\|RDREQ_sum\|duration_us" $OUT/summary_wg0.txt $OUT/summary_wg1.txt | head -60
