#!/bin/bash
TAG=${1:-r7k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 300 python scripts/diag_bench_timing.py > $OUT/diag_bench_timing.txt 2>&1; echo rc=$?; grep -v amdgpu.ids $OUT/diag_bench_timing.txt
timeout 300 python scripts/ab_step.py libganet_hip_new.so libganet_hip_new.so > $OUT/ab_step.txt 2>&1; tail -2 $OUT/ab_step.txt
