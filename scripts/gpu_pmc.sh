#!/bin/bash
# PMC counter passes (each its own rocprofv3 run, --pmc only with --kernel-trace): bash scripts/gpu_pmc.sh <tag> <stage>
TAG=${1:-pmc}; STAGE=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py $STAGE 2 > $OUT/p$i.log 2>&1
  echo "set $i rc=$?"
done
python $ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
