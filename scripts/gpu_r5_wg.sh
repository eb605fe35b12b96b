#!/bin/bash
# Round 5, first call: everything built after round 4's last GPU minute (emulator-verified, never run on a GPU, all off by default):
#   GANET_LGA_WG=1|2         all six LGA kernels of the step on ONE LDS ring per 256-thread workgroup (1: barrier per plane pair, 2: progress flags)
#   GANET_LGA_FG_FUSED=1     both filter-gradient passes of an LGA2's backward in one kernel (Python layer; ganet_lga2_filter_grad)
#   GANET_SGA_POINT_Q4=1     the SGA per-pixel gradient kernel on pixel quads (16-byte loads)
# Order: what cannot hang first -- (1) device parity of the barrier rings, the fused filter gradient, the quad kernel, each under a
# timeout of its own; (2) whole-step A/B on one box (hipGraph replay, interleaved: +-0.2 %); (3) the LGA / SGA kernels one by one;
# (4) fabric traffic of the step with the rings on (x over-fetch 1.5 - 1.9 x -> ?); and only then (5) the progress-flag form, which
# contains a poll loop: its parity tests and its A/B.  Variant libraries, if built, join (2) and (5):
#   python scripts/build_variants.py wg6:-DLGAP_WG_NR=6 wgf6:-DLGAP_WG_NR_FG=6 wg10:-DLGAP_WG_NR=10,-DLGAP_WG_NR_FG=8 \
#          wgs2:-DLGAP_WG_NR=10,-DLGAP_WG_SLACK=2,-DLGAP_WG_NR_FG=8 q4w5:-DGA_POINT_Q4_WAVES=5
#   (each ring variant checked under the emulator first: python scripts/sim_wg_variants.py <flags>)
# bash scripts/gpu_r5_wg.sh <tag>
TAG=${1:-r8a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
export GANET_TEST_WG=1
GANET_TEST_WG_FORMS=1 timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bounds.py tests/test_gpu_modules.py -x -q -m gpu \
  -k "workgroup_ring or lga_chain_on_end_aligned or paired or round5 or pixel_quads" > $OUT/tests_candidates.log 2>&1; echo "candidate tests (barrier rings, fused fg, quads) rc=$?"; tail -3 $OUT/tests_candidates.log
L=libganet_hip.so
LIBS="$L $L@GANET_SGA_POINT_Q4=1 $L@GANET_LGA_FG_FUSED=1 $L@GANET_LGA_WG=1 $L@GANET_LGA_WG=1,GANET_LGA_MIX=0 $L@GANET_LGA_WG=1,GANET_SGA_POINT_Q4=1"
[ -f ganet_amd/libganet_hip_q4w5.so ] && LIBS="$LIBS libganet_hip_q4w5.so@GANET_SGA_POINT_Q4=1"
for V in wg6 wgf6 wg10; do [ -f ganet_amd/libganet_hip_$V.so ] && LIBS="$LIBS libganet_hip_$V.so@GANET_LGA_WG=1"; done
timeout -k 5 400 python scripts/ab_step.py $LIBS > $OUT/ab_step.txt 2>&1; echo "ab_step rc=$?"; tail -14 $OUT/ab_step.txt
timeout -k 5 200 python scripts/ab_lga_stages.py $L $L@GANET_LGA_WG=1 > $OUT/ab_lga_stages.txt 2>&1; echo "ab_lga_stages rc=$?"; tail -6 $OUT/ab_lga_stages.txt
timeout -k 5 200 python scripts/ab_sga_stages.py $L $L@GANET_SGA_POINT_Q4=1 > $OUT/ab_sga_stages.txt 2>&1; echo "ab_sga_stages rc=$?"; tail -6 $OUT/ab_sga_stages.txt
for WG in 0 1; do
  ( cd /tmp; GANET_LGA_WG=$WG timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/pmc_wg$WG/p1 -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py step 3 > $OUT/pmc_wg$WG.log 2>&1; echo "pmc wg=$WG rc=$?" )
  python scripts/pmc_summary.py $OUT/pmc_wg$WG > $OUT/summary_wg$WG.txt 2>&1
  find $OUT/pmc_wg$WG -name '*.csv' -size +1M -delete
done
grep -h "^== lga\|RDREQ_sum\|duration_us" $OUT/summary_wg0.txt $OUT/summary_wg1.txt | head -60
# (5) the progress-flag form last
GANET_TEST_WG_FORMS=2 timeout -k 5 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py -x -q -m gpu -k "workgroup_ring or round5" > $OUT/tests_flags.log 2>&1; echo "flag-ring tests rc=$?"; tail -3 $OUT/tests_flags.log
LIBS="$L $L@GANET_LGA_WG=1 $L@GANET_LGA_WG=2 $L@GANET_LGA_WG=2,GANET_LGA_SEGS=2"
for V in wg6 wg10 wgs2; do [ -f ganet_amd/libganet_hip_$V.so ] && LIBS="$LIBS libganet_hip_$V.so@GANET_LGA_WG=2"; done
timeout -k 5 300 python scripts/ab_step.py $LIBS > $OUT/ab_step_flags.txt 2>&1; echo "ab_step (flags) rc=$?"; tail -10 $OUT/ab_step_flags.txt
