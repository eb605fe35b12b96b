#!/bin/bash
TAG=${1:-r7e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python scripts/ab_sga_stages.py libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=1 libganet_hip_new.so@GANET_SGA_TILED=2 libganet_hip_new.so@GANET_SGA_TILED=3 > $OUT/ab_sga_stages.txt 2>&1; echo "rc=$?"; tail -8 $OUT/ab_sga_stages.txt
