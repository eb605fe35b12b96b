#!/bin/bash
# round 4, session 2: op tests (incl. the tiled private workspace), whole-step A/B of GANET_SGA_TILED, stage timings, bench
TAG=${1:-r7b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
{ nproc; lscpu | grep -m1 'Model name'; rocminfo | grep -m3 -E 'Marketing Name|gfx'; } > $OUT/host.txt 2>&1
echo "== pytest -m gpu (ops)"; SECONDS=0
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_model.py > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -4 $OUT/pytest_gpu.txt
echo "== whole-step A/B"; SECONDS=0
timeout 600 python scripts/ab_step.py libganet_hip_r3.so libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=1 libganet_hip_new.so@GANET_SGA_TILED=2 libganet_hip_new.so@GANET_SGA_TILED=3 libganet_hip_r3.so libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=3 > $OUT/ab_step.txt 2>&1; echo "rc=$? (${SECONDS}s)"; cat $OUT/ab_step.txt | tail -9
echo "== LGA stage timings"; SECONDS=0
timeout 600 python scripts/ab_lga_stages.py libganet_hip_r3.so libganet_hip_new.so > $OUT/ab_lga_stages.txt 2>&1; echo "rc=$? (${SECONDS}s)"; tail -4 $OUT/ab_lga_stages.txt
echo "== SGA stage timings"; SECONDS=0
timeout 600 python scripts/ab_sga_stages.py libganet_hip_new.so libganet_hip_new.so@GANET_SGA_TILED=1 libganet_hip_new.so@GANET_SGA_TILED=3 > $OUT/ab_sga_stages.txt 2>&1; echo "rc=$? (${SECONDS}s)"; tail -6 $OUT/ab_sga_stages.txt
echo "== bench"; SECONDS=0
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? (${SECONDS}s)"; cut -c1-300 $OUT/bench.json; tail -2 $OUT/bench.err | cut -c1-200
GANET_SGA_TILED=3 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_tiled3.json 2> $OUT/bench_tiled3.err; echo "bench(tiled=3) rc=$?"; cut -c1-300 $OUT/bench_tiled3.json
echo "== done"
