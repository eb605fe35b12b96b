#!/bin/bash
# same-box re-sweep of compile-time parameters tuned in earlier rounds, on this round's code (whole step, hipGraph A/B)
TAG=${1:-r7u}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
L=""; for v in new la1 nr4 mdu1 mdu3 pdu1 pdu3 new la1 nr4 mdu1 mdu3 pdu1 pdu3; do L="$L libganet_hip_$v.so"; done
timeout 600 python scripts/ab_step.py $L > $OUT/ab_step.txt 2>&1; echo rc=$?; tail -14 $OUT/ab_step.txt
