#!/bin/bash
# One GPU session.  gpurun --timeout N -- 'bash scripts/gpu_session.sh <tag> [parts]'
#   parts: any of  ubench tests smoke bench prof pmc pmcwave modeltests model nofind census ab stages  (default: tests smoke bench prof model)
#   pmc: memory-side passes over the step -> traffic_pmc.json;  pmcwave: wave-level issue / wait / LDS / MFMA counter sets over
#   scripts/prof_stage.py $STAGE (default: step);  census: the fused-vs-stock op censuses (scripts/bench_census*.py,
#   bench_residual_tail.py);  ab: scripts/ab_step.py $AB_ARGS (same-box whole-step A/B of library builds / options);
#   stages: scripts/ab_lga_stages.py + ab_sga_stages.py $AB_ARGS (every kernel of the step in sequence, per library)
# (The one-session scripts of rounds 4-5 -- gpu_s1..s8, gpu_pmc*, gpu_diag*, gpu_sweep, gpu_ab -- are these parts now; what each
# of them measured is in profiles/ under its tag.)
# Everything worth keeping goes to gpurun_out/<tag>/ (merged back into the dev container).
TAG=${1:-r2}
PARTS=${2:-"tests smoke bench prof model"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT $OUT/pmc
cd $ROOT
export TMPDIR=/tmp
# MIOpen: no gfx950 find-db / kernel-db ships with the image; keep what this box compiles and bring back what an
# earlier call compiled (miopen_cache/ travels with the snapshot, gpurun_out/ does not)
mkdir -p /tmp/miopen/db /tmp/miopen/cache
[ -d $ROOT/miopen_cache ] && cp -r $ROOT/miopen_cache/. /tmp/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache
has() { [[ " $PARTS " == *" $1 "* ]]; }
{
  echo "== host"; nproc; lscpu | grep -m1 'Model name'; rocminfo | grep -m3 -E 'Marketing Name|gfx'
} > $OUT/host.txt 2>&1
if has ubench; then
  echo "== ubench"
  for b in scripts/ubench/*.bin; do [ -x $b ] && { echo "-- $b"; timeout 120 $b; } ; done > $OUT/ubench.txt 2>&1
  tail -30 $OUT/ubench.txt
fi
if has tests; then
  echo "== pytest -m gpu (ops)"
  SECONDS=0
  timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_model.py > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -25 $OUT/pytest_gpu.txt
fi
if has smoke; then
  echo "== smoke"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.txt
fi
if has bench; then
  echo "== bench"
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
fi
if has prof; then
  echo "== rocprof kernel trace (bench)"
  ( cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" )
  find $OUT/prof -name '*kernel_stats*' | head -1 | xargs -r head -16
fi
if has pmc; then
  echo "== memory-side PMC passes (traffic per kernel at the L2 <-> fabric boundary)"
  ( cd /tmp; i=0
    for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
               "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum" \
               "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc/p$i -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py all 2 > $OUT/pmc/p$i.log 2>&1
      echo "pmc set $i rc=$?"
    done )
  python scripts/pmc_summary.py $OUT/pmc > $OUT/pmc/summary.txt 2>&1
  python scripts/pmc_traffic.py $OUT/pmc/summary.txt $OUT/traffic_pmc.json > /dev/null 2>&1; echo "traffic rc=$?"
  find $OUT/pmc -name '*.csv' -size +2M -delete 2>/dev/null
fi
if has census; then
  echo "== census (GA work of one inference pass / one training step, stock statements vs fused modules)"
  timeout 300 python scripts/bench_residual_tail.py > $OUT/bench_residual_tail.json 2> $OUT/bench_residual_tail.err; echo "tail rc=$?"; cat $OUT/bench_residual_tail.json
  timeout 300 python scripts/bench_census_infer.py --no-tail > $OUT/bench_census_infer_no_tail.json 2>> $OUT/census.err; echo "rc=$?"; cat $OUT/bench_census_infer_no_tail.json
  timeout 300 python scripts/bench_census_infer.py > $OUT/bench_census_infer.json 2>> $OUT/census.err; echo "rc=$?"; cat $OUT/bench_census_infer.json
  timeout 300 python scripts/bench_census.py > $OUT/bench_census.json 2>> $OUT/census.err; echo "rc=$?"; cat $OUT/bench_census.json
fi
if has ab; then
  echo "== same-box whole-step A/B: ab_step.py $AB_ARGS"
  timeout 900 python scripts/ab_step.py $AB_ARGS > $OUT/ab_step.txt 2> $OUT/ab_step.err; echo "ab rc=$?"; cat $OUT/ab_step.txt; tail -3 $OUT/ab_step.err
fi
if has pmcwave; then
  echo "== wave-level counter passes over prof_stage.py ${STAGE:-step} (at most four TCC / TCP counters per pass: larger sets hung in round 3)"
  mkdir -p $OUT/pmcwave
  ( cd /tmp; i=0
    for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
               "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
               "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmcwave/p$i -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py ${STAGE:-step} 3 > $OUT/pmcwave/p$i.log 2>&1
      echo "pmcwave set $i rc=$?"
    done )
  python scripts/pmc_summary.py $OUT/pmcwave > $OUT/pmcwave/summary.txt 2>&1
  find $OUT/pmcwave -name '*.csv' -size +2M -delete 2>/dev/null
  grep -c "^==" $OUT/pmcwave/summary.txt
fi
if has stages; then
  echo "== every kernel of the step in sequence, per library: $AB_ARGS"
  timeout 600 python scripts/ab_lga_stages.py $AB_ARGS > $OUT/ab_lga_stages.txt 2>&1; echo "lga stages rc=$?"; grep -v amdgpu.ids $OUT/ab_lga_stages.txt | cut -c1-400
  timeout 600 python scripts/ab_sga_stages.py $AB_ARGS > $OUT/ab_sga_stages.txt 2>&1; echo "sga stages rc=$?"; grep -v amdgpu.ids $OUT/ab_sga_stages.txt | cut -c1-400
fi
if has modeltests; then
  echo "== model tests (reference models on the drop-in, GPU vs CPU-oracle twin)"
  SECONDS=0
  timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s > $OUT/pytest_model.txt 2>&1; echo "model pytest rc=$? (${SECONDS}s)"; grep -E "gpu vs|fused vs|MODE_B|passed|failed" $OUT/pytest_model.txt
fi
if has model; then
  echo "== cfg3 inference (GANet_deep 1248x384)"
  SECONDS=0
  timeout 600 python -m harness.infer --kernel_share > $OUT/infer_stock.json 2> $OUT/infer_stock.err; echo "infer rc=$? (${SECONDS}s)"; cat $OUT/infer_stock.json; tail -3 $OUT/infer_stock.err
  timeout 400 python -m harness.infer --fused --kernel_share > $OUT/infer_fused.json 2> $OUT/infer_fused.err; echo "infer fused rc=$?"; cat $OUT/infer_fused.json; tail -3 $OUT/infer_fused.err
  echo "== cfg4 per-GPU training step (GANet_deep 240x624, 1 sample)"
  SECONDS=0
  timeout 600 python -m harness.train --steps 5 --warmup 2 --kernel_share > $OUT/train_stock.json 2> $OUT/train_stock.err; echo "train rc=$? (${SECONDS}s)"; cat $OUT/train_stock.json; tail -3 $OUT/train_stock.err
  timeout 400 python -m harness.train --steps 5 --warmup 2 --fused --kernel_share > $OUT/train_fused.json 2> $OUT/train_fused.err; echo "train fused rc=$?"; cat $OUT/train_fused.json; tail -3 $OUT/train_fused.err
fi
if has nofind; then
  echo "== MIOpen immediate mode (PyTorch default) for comparison"
  timeout 400 python -m harness.infer --no_miopen_find > $OUT/infer_nofind.json 2> $OUT/infer_nofind.err; echo "rc=$?"; cat $OUT/infer_nofind.json
  timeout 400 python -m harness.train --no_miopen_find --steps 3 --warmup 1 > $OUT/train_nofind.json 2> $OUT/train_nofind.err; echo "rc=$?"; cat $OUT/train_nofind.json
fi
# keep the MIOpen products for the next call if they are small enough to come back
du -sh /tmp/miopen 2>/dev/null
if [ "$(du -sm /tmp/miopen | cut -f1)" -lt 40 ]; then mkdir -p $OUT/miopen_cache && cp -r /tmp/miopen/. $OUT/miopen_cache/; fi
# trim the raw traces (only the stats travel well)
find $OUT -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
echo "== done"
