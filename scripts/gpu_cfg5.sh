#!/bin/bash
# cfg5 per-GPU training step (GANet_deep, 2 x 960x528) in MIOpen FIND mode, problem by problem (round 5:
# 103 problems, all found in <= 9.3 s each, step 520 ms -- profiles/r8g_*, r8h_*).  The one script for this shape: the bounded
# in-process attempts of round 4 (gpu_cfg5.sh / gpu_cfg5c.sh of that round) and the separate step script are folded in here.
# Rounds 2-4: a cold find pass inside the training process outlasted every budget, and a killed process loses what it found
# (MIOpen writes its find records at process exit).  Here every convolution problem of the step gets a short-lived process of its
# own -- MIOpenDriver with the problem's logged command line (its Find call, no CPU verification) -- under its own timeout, so a kill loses ONE record and
# the slow problems are named with their times; the training step then runs in find mode on the find-db that is left.
#   bash scripts/gpu_cfg5.sh <tag> [per-problem seconds] [total seconds for the problem loop] [seconds for the step]
TAG=${1:-r8e}
PER=${2:-60}
TOTAL=${3:-800}
STEP_BUDGET=${4:-300}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
mkdir -p /tmp/miopen/db /tmp/miopen/cache
[ -d $ROOT/miopen_cache ] && cp -r $ROOT/miopen_cache/. /tmp/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache
export MIOPEN_DEBUG_CONV_GEMM=0 MIOPEN_DEBUG_CONV_FFT=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
save_db() { if [ "$(du -sm /tmp/miopen | cut -f1)" -lt 48 ]; then mkdir -p $OUT/miopen_cache && cp -r /tmp/miopen/. $OUT/miopen_cache/ 2>/dev/null; fi; }
# (1) the step's complete problem list: one step in immediate mode (finishes: 519 ms per step in round 4) with command logging on
SECONDS=0
MIOPEN_ENABLE_LOGGING_CMD=1 timeout -k 5 400 python -m harness.train --crop_height 528 --crop_width 960 --batch 2 --steps 1 --warmup 1 --fused --no_miopen_find \
  > $OUT/cfg5_immediate.json 2> $OUT/cfg5_immediate.err
echo "immediate-mode step rc=$? (${SECONDS}s)"; cut -c1-300 $OUT/cfg5_immediate.json
grep "LogCmdConvolution" $OUT/cfg5_immediate.err | sed 's/.*MIOpenDriver //' | sort -u > $OUT/problems.txt
rm -f $OUT/cfg5_immediate.err
echo "problems: $(wc -l < $OUT/problems.txt)"
save_db
# (2) find, one process per problem
: > $OUT/find_times.txt
T0=$(date +%s)
N=0; DONE=0; SLOW=0
while read -r CMD; do
  N=$((N+1))
  if [ $(( $(date +%s) - T0 )) -gt $TOTAL ]; then echo "NOT_REACHED $CMD" >> $OUT/find_times.txt; continue; fi
  S=$(date +%s.%N)
  timeout -k 3 $PER /opt/rocm/bin/MIOpenDriver $CMD -V 0 -i 1 < /dev/null > $OUT/driver_last.log 2>&1
  RC=$?
  E=$(python3 -c "import time,sys; print(f'{time.time()-float(sys.argv[1]):.1f}')" $S)
  if [ $RC -eq 0 ]; then DONE=$((DONE+1)); echo "OK $E s  $CMD" >> $OUT/find_times.txt
  else SLOW=$((SLOW+1)); echo "RC=$RC after $E s  $CMD" >> $OUT/find_times.txt; fi
  [ $((N % 10)) -eq 0 ] && save_db
done < $OUT/problems.txt
echo "find loop: $DONE finished, $SLOW cut off / failed, of $N ($(( $(date +%s) - T0 )) s)"
save_db
sort -k2 -n -r $OUT/find_times.txt | grep "^OK" | head -5
grep -v "^OK" $OUT/find_times.txt | head -20 | cut -c1-220
wc -l /tmp/miopen/db/*ufdb.txt 2>/dev/null | tail -1
# (3) the training step in find mode on what is there now.  This image carries two MIOpen builds (3.5.0 inside the torch wheel,
# 3.5.1 under /opt/rocm = what MIOpenDriver links) with one user find-db file per version: merge them first
python scripts/miopen_merge_ufdb.py /tmp/miopen/db
SECONDS=0
timeout -k 5 -s ABRT $STEP_BUDGET python -X faulthandler -m harness.train --crop_height 528 --crop_width 960 --batch 2 --steps 2 --warmup 1 --fused > $OUT/cfg5_train_find.json 2> $OUT/cfg5_train_find.err
echo "find-mode step rc=$? (${SECONDS}s)"; cut -c1-600 $OUT/cfg5_train_find.json; tail -n 3 $OUT/cfg5_train_find.err | cut -c1-300
save_db
echo "== done"
