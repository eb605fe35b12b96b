#!/bin/bash
# generic same-box whole-step A/B:  bash scripts/gpu_ab.sh <tag> "<ab_step.py arguments>"  -> gpurun_out/<tag>/ab_step.txt
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
{ rocminfo | grep -m1 -E 'Marketing Name'; } > $OUT/host.txt 2>&1
timeout -k 5 600 python scripts/ab_step.py $2 > $OUT/ab_step.txt 2>&1; echo "ab_step rc=$?"; grep median $OUT/ab_step.txt
