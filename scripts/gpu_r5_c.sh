#!/bin/bash
# Round 5, call 3: the tree after the candidates were decided (workgroup rings on by default, quads / fused filter gradient / flag
# rings deleted): (1) the LGA + SGA op tests on the device, (2) whole-step A/B of the filter gradient's ring depth (LGAP_WG_NR_FG
# 5 .. 8) and of the rings against the one-wave fallback, (3) bench.py.
TAG=${1:-r8c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_model.py > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
L=libganet_hip.so
LIBS="$L $L@GANET_LGA_WG=0"
for V in wgf5 wgf6 wgf7; do [ -f ganet_amd/libganet_hip_$V.so ] && LIBS="$LIBS libganet_hip_$V.so"; done
timeout -k 5 400 python scripts/ab_step.py $LIBS $L > $OUT/ab_step.txt 2>&1; echo "ab_step rc=$?"; tail -8 $OUT/ab_step.txt
timeout -k 5 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
