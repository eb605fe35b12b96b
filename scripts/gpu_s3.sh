#!/bin/bash
# round 4, session 3: the whole validation set on one box -- op + model tests, smoke, bench, kernel trace of the bench command,
# PMC passes over the step's 16 launches (traffic at the L2 <-> fabric boundary, wave-level issue / wait counters)
TAG=${1:-r7f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT $OUT/pmc
cd $ROOT
export TMPDIR=/tmp
mkdir -p /tmp/miopen/db /tmp/miopen/cache
[ -d $ROOT/miopen_cache ] && cp -r $ROOT/miopen_cache/. /tmp/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache
{ nproc; lscpu | grep -m1 'Model name'; rocminfo | grep -m3 -E 'Marketing Name|gfx'; } > $OUT/host.txt 2>&1
echo "== pytest -m gpu (ops)"; SECONDS=0
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_model.py > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -3 $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
echo "== bench"; SECONDS=0
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? (${SECONDS}s)"; cut -c1-260 $OUT/bench.json
echo "== rocprof kernel trace (bench)"
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" )
find $OUT/prof -name '*kernel_stats*' | head -1 | xargs -r head -18 | cut -c1-150
echo "== PMC passes over the step"
( cd /tmp; i=0
  for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc/p$i -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py step 3 > $OUT/pmc/p$i.log 2>&1
    echo "pmc set $i rc=$?"
  done )
python scripts/pmc_summary.py $OUT/pmc > $OUT/pmc/summary.txt 2>&1
python scripts/pmc_traffic.py $OUT/pmc/summary.txt $OUT/traffic_pmc.json > /dev/null 2>&1; echo "traffic rc=$?"
find $OUT/pmc -name '*.csv' -size +2M -delete 2>/dev/null
echo "== model tests"; SECONDS=0
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s > $OUT/pytest_model.txt 2>&1; echo "model pytest rc=$? (${SECONDS}s)"; grep -E "passed|failed" $OUT/pytest_model.txt
find $OUT -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
du -sh /tmp/miopen 2>/dev/null
if [ "$(du -sm /tmp/miopen | cut -f1)" -lt 40 ]; then mkdir -p $OUT/miopen_cache && cp -r /tmp/miopen/. $OUT/miopen_cache/; fi
echo "== done"
