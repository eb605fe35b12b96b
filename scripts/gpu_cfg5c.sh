#!/bin/bash
# cfg5 (and, for calibration, cfg4) training step in MIOpen IMMEDIATE mode with the naive-direct / GEMM / FFT solver families
# disabled: the first applicable solver is then a CK implicit-GEMM / Winograd / direct-asm kernel, one compilation per problem,
# no benchmarking -- 14 s for the first cfg5 step where find mode outlasted 1,680 s (profiles/r7v_*)
TAG=${1:-r7w}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
mkdir -p /tmp/miopen/db /tmp/miopen/cache
[ -d $ROOT/miopen_cache ] && cp -r $ROOT/miopen_cache/. /tmp/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache
export MIOPEN_DEBUG_CONV_GEMM=0 MIOPEN_DEBUG_CONV_FFT=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
SECONDS=0
timeout -k 5 300 python -m harness.train --crop_height 528 --crop_width 960 --batch 2 --steps 5 --warmup 2 --fused --no_miopen_find --kernel_share > $OUT/cfg5_train_fused_immediate.json 2> $OUT/cfg5_fused.err; echo "cfg5 fused rc=$? (${SECONDS}s)"; cut -c1-420 $OUT/cfg5_train_fused_immediate.json
timeout -k 5 300 python -m harness.train --crop_height 528 --crop_width 960 --batch 2 --steps 5 --warmup 2 --no_miopen_find --kernel_share > $OUT/cfg5_train_stock_immediate.json 2> $OUT/cfg5_stock.err; echo "cfg5 stock rc=$? (${SECONDS}s)"; cut -c1-420 $OUT/cfg5_train_stock_immediate.json
timeout -k 5 300 python -m harness.train --steps 5 --warmup 2 --fused --no_miopen_find > $OUT/cfg4_train_fused_immediate.json 2> $OUT/cfg4_fused.err; echo "cfg4 fused (immediate, same env) rc=$? (${SECONDS}s)"; cut -c1-300 $OUT/cfg4_train_fused_immediate.json
tail -2 $OUT/cfg5_fused.err | cut -c1-200
