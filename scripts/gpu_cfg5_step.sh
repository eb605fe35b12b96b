#!/bin/bash
# the cfg5 per-GPU training step in MIOpen find mode on the find-db carried in miopen_cache/ (after scripts/gpu_cfg5_by_problem.sh
# left MIOpenDriver's per-problem records there): merge the per-version find-dbs (scripts/miopen_merge_ufdb.py), then the step.
#   bash scripts/gpu_cfg5_step.sh <tag> [seconds]
TAG=${1:-r8h}
BUDGET=${2:-400}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
mkdir -p /tmp/miopen/db /tmp/miopen/cache
[ -d $ROOT/miopen_cache ] && cp -r $ROOT/miopen_cache/. /tmp/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache
export MIOPEN_DEBUG_CONV_GEMM=0 MIOPEN_DEBUG_CONV_FFT=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
python scripts/miopen_merge_ufdb.py /tmp/miopen/db
save_db() { if [ "$(du -sm /tmp/miopen | cut -f1)" -lt 48 ]; then mkdir -p $OUT/miopen_cache && cp -r /tmp/miopen/. $OUT/miopen_cache/ 2>/dev/null; fi; }
( while sleep 50; do save_db; done ) &
COPIER=$!
SECONDS=0
MIOPEN_ENABLE_LOGGING_CMD=1 timeout -k 5 -s ABRT $BUDGET python -X faulthandler -m harness.train --crop_height 528 --crop_width 960 --batch 2 --steps 2 --warmup 1 --fused > $OUT/cfg5_train_find.json 2> $OUT/cfg5_train_find.err
echo "find-mode step rc=$? (${SECONDS}s)"; cut -c1-700 $OUT/cfg5_train_find.json
kill $COPIER 2>/dev/null
echo "conv problems logged: $(grep -c LogCmdConvolution $OUT/cfg5_train_find.err)  find calls logged: $(grep -c -i 'LogCmdFindConvolution' $OUT/cfg5_train_find.err)"
grep -v "LogCmd" $OUT/cfg5_train_find.err | tail -n 40 | cut -c1-200 > $OUT/cfg5_train_find.tail.txt; tail -n 30 $OUT/cfg5_train_find.tail.txt
grep "LogCmdConvolution" $OUT/cfg5_train_find.err | sed 's/.*MIOpenDriver //' | tail -n 3
rm -f $OUT/cfg5_train_find.err
save_db
echo "== done"
