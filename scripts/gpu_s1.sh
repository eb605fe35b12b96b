#!/bin/bash
# round 4, session 1: op tests on the new default library, whole-step A/B of the LGA instruction-diet variants against round 3's
# library, LGA / SGA stage timings of the timing-only ablations, one bench line
TAG=${1:-r7a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
{ nproc; lscpu | grep -m1 'Model name'; rocminfo | grep -m3 -E 'Marketing Name|gfx'; } > $OUT/host.txt 2>&1
echo "== pytest -m gpu (ops)"; SECONDS=0
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_model.py > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -4 $OUT/pytest_gpu.txt
echo "== whole-step A/B"; SECONDS=0
timeout 600 python scripts/ab_step.py libganet_hip_r3.so libganet_hip_new.so libganet_hip_la1.so libganet_hip_m0save.so libganet_hip_norow.so libganet_hip_r3.so libganet_hip_new.so > $OUT/ab_step.txt 2>&1; echo "rc=$? (${SECONDS}s)"; cat $OUT/ab_step.txt | tail -8
echo "== LGA stage timings"; SECONDS=0
timeout 600 python scripts/ab_lga_stages.py libganet_hip_r3.so libganet_hip_new.so libganet_hip_la1.so libganet_hip_notaps.so > $OUT/ab_lga_stages.txt 2>&1; echo "rc=$? (${SECONDS}s)"; tail -8 $OUT/ab_lga_stages.txt
echo "== SGA stage timings (tiled-write ablation)"; SECONDS=0
timeout 600 python scripts/ab_sga_stages.py libganet_hip_new.so libganet_hip_tiledw.so > $OUT/ab_sga_stages.txt 2>&1; echo "rc=$? (${SECONDS}s)"; tail -4 $OUT/ab_sga_stages.txt
echo "== bench"; SECONDS=0
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? (${SECONDS}s)"; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== done"
