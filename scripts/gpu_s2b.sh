#!/bin/bash
TAG=${1:-r7c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== tiled tests"; SECONDS=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiled and shape0" > $OUT/pytest_tiled.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -2 $OUT/pytest_tiled.txt
echo "== whole-step A/B"; SECONDS=0
timeout 600 python scripts/ab_step.py libganet_hip_r3.so libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=1 libganet_hip_new.so@GANET_SGA_TILED=2 libganet_hip_new.so@GANET_SGA_TILED=3 libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=2 libganet_hip_new.so@GANET_SGA_TILED=3 > $OUT/ab_step.txt 2>&1; echo "rc=$? (${SECONDS}s)"; cat $OUT/ab_step.txt | tail -9
echo "== SGA stage timings"; SECONDS=0
timeout 600 python scripts/ab_sga_stages.py libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=1 libganet_hip_new.so@GANET_SGA_TILED=3 > $OUT/ab_sga_stages.txt 2>&1; echo "rc=$? (${SECONDS}s)"; tail -6 $OUT/ab_sga_stages.txt
echo "== done"
