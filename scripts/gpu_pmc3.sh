#!/bin/bash
# wave-level PMC passes (memory side: scripts/gpu_pmc2.sh -- at most FOUR TCC / TCP counters per pass: passes with six or seven
# of them hung until the timeout in round 3 and cost 15 GPU-minutes) for ONE stage: bash scripts/gpu_pmc3.sh <tag> <stage>      (stage: see scripts/prof_stage.py)
TAG=${1:-pmc3}; STAGE=${2:-sga_fwd}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py $STAGE 2 > $OUT/p$i.log 2>&1
  echo "set $i rc=$?"; tail -2 $OUT/p$i.log | grep -i error
done
python $ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/p*/
