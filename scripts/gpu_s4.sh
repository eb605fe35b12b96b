#!/bin/bash
# round 4, session 4: model-level numbers on this round's kernels (cfg3 inference, cfg4 per-GPU training step, stock + fused),
# then one bounded attempt at the cfg5 training step whose MIOpen products (compiled kernels, find results) come back
# whatever happens, so that the next attempt starts where this one stopped
TAG=${1:-r7g}
BUDGET=${2:-420}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
mkdir -p /tmp/miopen/db /tmp/miopen/cache
[ -d $ROOT/miopen_cache ] && cp -r $ROOT/miopen_cache/. /tmp/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache
if [ "$3" != "cfg5only" ]; then
echo "== cfg3 inference (GANet_deep 1248x384)"; SECONDS=0
timeout 600 python -m harness.infer --kernel_share > $OUT/infer_stock.json 2> $OUT/infer_stock.err; echo "infer rc=$? (${SECONDS}s)"; cut -c1-400 $OUT/infer_stock.json
timeout 400 python -m harness.infer --fused --kernel_share > $OUT/infer_fused.json 2> $OUT/infer_fused.err; echo "infer fused rc=$?"; cut -c1-400 $OUT/infer_fused.json
echo "== cfg4 per-GPU training step (GANet_deep 240x624, 1 sample)"; SECONDS=0
timeout 700 python -m harness.train --steps 5 --warmup 2 --kernel_share > $OUT/train_stock.json 2> $OUT/train_stock.err; echo "train rc=$? (${SECONDS}s)"; cut -c1-400 $OUT/train_stock.json
timeout 500 python -m harness.train --steps 5 --warmup 2 --fused --kernel_share > $OUT/train_fused.json 2> $OUT/train_fused.err; echo "train fused rc=$?"; cut -c1-400 $OUT/train_fused.json
fi
echo "== cfg5 per-GPU training step (GANet_deep 2 x 960x528), at most ${BUDGET}s"; SECONDS=0
timeout -s INT $BUDGET python -m harness.train --crop_height 528 --crop_width 960 --batch 2 --steps 2 --warmup 1 --fused > $OUT/cfg5_train.json 2> $OUT/cfg5_train.err; echo "cfg5 rc=$? (${SECONDS}s)"; cut -c1-500 $OUT/cfg5_train.json; tail -3 $OUT/cfg5_train.err | cut -c1-200
du -sh /tmp/miopen /tmp/miopen/* 2>/dev/null
if [ "$(du -sm /tmp/miopen | cut -f1)" -lt 48 ]; then mkdir -p $OUT/miopen_cache && cp -r /tmp/miopen/. $OUT/miopen_cache/; else echo "miopen products too large to bring back"; fi
echo "== done"
