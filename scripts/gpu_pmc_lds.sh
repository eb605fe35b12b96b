#!/bin/bash
# LDS- and MFMA-side counters of the step's kernels (one rocprofv3 --pmc pass per set, --kernel-trace only):
#   how busy is the LDS array under the LGA kernels (SQ_LDS_IDX_ACTIVE, bank / address conflicts, unaligned stalls), what share of
#   the instructions are LDS instructions, and the MFMA pipe's busy cycles (expected 0: the kernels issue no MFMA).
#   bash scripts/gpu_pmc_lds.sh <tag>
TAG=${1:-r8q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT/pmc; cd /tmp
export TMPDIR=/tmp
i=0
for SET in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc/p$i -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py step 3 > $OUT/pmc/p$i.log 2>&1
  echo "pmc set $i rc=$?"; tail -n 2 $OUT/pmc/p$i.log | cut -c1-200
done
cd $ROOT
python scripts/pmc_summary.py $OUT/pmc > $OUT/summary.txt 2>&1
find $OUT/pmc -name '*.csv' -size +2M -delete 2>/dev/null
grep -c "^==" $OUT/summary.txt
