#!/bin/bash
# One BOUNDED attempt at the cfg5 per-GPU training step (GANet_deep, 2 x 960x528, fused call sites, MIOpen find mode).
# Rounds 2-4 learnt the hard way that a cold find pass for this set of shapes outlasts every budget tried (200 s, 1,000 s,
# 1,680 s), that SIGINT does not stop a process inside MIOpen's kernel compilation, and that what a killed call has compiled is
# lost unless it is copied out while the call runs.  So: the naive direct / GEMM / FFT solver families are taken out of the
# search (their benchmarks alone take seconds per shape at this size; the winners at the cfg4 shapes were CK implicit-GEMM and
# Winograd kernels), MIOpen's user db + kernel cache are copied to gpurun_out/ once a minute, and the step is killed hard.
TAG=${1:-r7h}
BUDGET=${2:-780}
MODE=${3:-find}      # find: MIOpen find mode (times the applicable solvers per problem); immediate: PyTorch's default mode, where MIOpen
                     # takes the first applicable solver -- with the naive / GEMM / FFT families disabled below that is a CK implicit-GEMM,
                     # Winograd or direct-asm kernel, one compilation per problem and no benchmarking
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
mkdir -p /tmp/miopen/db /tmp/miopen/cache
[ -d $ROOT/miopen_cache ] && cp -r $ROOT/miopen_cache/. /tmp/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache
export MIOPEN_DEBUG_CONV_GEMM=0 MIOPEN_DEBUG_CONV_FFT=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
export MIOPEN_ENABLE_LOGGING_CMD=1
( while sleep 45; do
    if [ "$(du -sm /tmp/miopen | cut -f1)" -lt 40 ]; then mkdir -p $OUT/miopen_cache && cp -r /tmp/miopen/. $OUT/miopen_cache/ 2>/dev/null; fi
    grep -c "MIOpenDriver" $OUT/cfg5_train.err > $OUT/progress_conv_cmds.txt 2>/dev/null
  done ) &
COPIER=$!
SECONDS=0
EXTRA=""; [ "$MODE" = "immediate" ] && EXTRA="--no_miopen_find"
timeout -k 5 -s INT $BUDGET python -m harness.train --crop_height 528 --crop_width 960 --batch 2 --steps 2 --warmup 1 --fused $EXTRA > $OUT/cfg5_train.json 2> $OUT/cfg5_train.err
echo "cfg5 rc=$? (${SECONDS}s)"; cut -c1-600 $OUT/cfg5_train.json
kill $COPIER 2>/dev/null
grep -c "MIOpenDriver" $OUT/cfg5_train.err; grep "MIOpenDriver" $OUT/cfg5_train.err | sort -u | wc -l
grep -v "MIOpenDriver" $OUT/cfg5_train.err | tail -5 | cut -c1-300
du -sh /tmp/miopen /tmp/miopen/* 2>/dev/null
if [ "$(du -sm /tmp/miopen | cut -f1)" -lt 40 ]; then mkdir -p $OUT/miopen_cache && cp -r /tmp/miopen/. $OUT/miopen_cache/; fi
# keep the log small
grep "MIOpenDriver" $OUT/cfg5_train.err | sort -u > $OUT/cfg5_conv_shapes.txt; grep -v "MIOpenDriver" $OUT/cfg5_train.err | tail -50 > $OUT/cfg5_train.tail.err; rm -f $OUT/cfg5_train.err
echo "== done"
