#!/bin/bash
TAG=${1:-r7m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
echo "== LGA tests"; SECONDS=0
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bounds.py tests/test_gpu_modules.py -m gpu -q -x -k "lga or Lga or LGA" > $OUT/pytest_lga.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -2 $OUT/pytest_lga.txt
timeout 600 python scripts/ab_step.py libganet_hip_base5.so libganet_hip_new.so libganet_hip_base5.so libganet_hip_new.so libganet_hip_base5.so libganet_hip_new.so > $OUT/ab_step.txt 2>&1; echo rc=$?; tail -6 $OUT/ab_step.txt
timeout 600 python scripts/ab_lga_stages.py libganet_hip_base5.so libganet_hip_new.so > $OUT/ab_lga_stages.txt 2>&1; tail -2 $OUT/ab_lga_stages.txt | cut -c1-330
