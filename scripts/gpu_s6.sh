#!/bin/bash
TAG=${1:-r7l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 600 python scripts/ab_step.py libganet_hip_new.so libganet_hip_new.so libganet_hip_pxcd.so libganet_hip_la1.so libganet_hip_new.so libganet_hip_pxcd.so libganet_hip_la1.so > $OUT/ab_step.txt 2>&1; echo rc=$?; tail -7 $OUT/ab_step.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-230 $OUT/bench.json
