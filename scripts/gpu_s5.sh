#!/bin/bash
TAG=${1:-r7i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== LGA / module tests"; SECONDS=0
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py tests/test_gpu_bounds.py -m gpu -q -x -k "Lga2 or lga2 or modules_autograd" > $OUT/pytest_lga.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -2 $OUT/pytest_lga.txt
echo "== whole-step A/B (base4 = before the edge-sum side channel)"; SECONDS=0
timeout 600 python scripts/ab_step.py libganet_hip_new.so@GANET_LGA_EDGES=0 libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=0 libganet_hip_new.so@GANET_LGA_EDGES=0 libganet_hip_new.so > $OUT/ab_step.txt 2>&1; echo "rc=$? (${SECONDS}s)"; tail -4 $OUT/ab_step.txt
echo "== LGA stage timings"
timeout 600 python scripts/ab_lga_stages.py libganet_hip_new.so > $OUT/ab_lga_stages.txt 2>&1; echo "rc=$?"; tail -2 $OUT/ab_lga_stages.txt | cut -c1-400
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench.json
echo "== done"
