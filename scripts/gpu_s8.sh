#!/bin/bash
# traffic of the per-pixel kernel with and without the XCD-aware block order (does the half-line over-fetch go away? does time follow?)
TAG=${1:-r7r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for L in new pxcd; do
  ( cd /tmp; GANET_PROF_LIB=libganet_hip_$L.so timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/pmc_$L/p1 -o pmc --output-format csv -- python $ROOT/scripts/prof_stage.py sga_bwd 3 > $OUT/pmc_$L.log 2>&1; echo "pmc $L rc=$?" )
  python scripts/pmc_summary.py $OUT/pmc_$L > $OUT/summary_$L.txt 2>&1
  grep -A8 "sga_bwd_point" $OUT/summary_$L.txt | head -9
  find $OUT/pmc_$L -name '*.csv' -size +1M -delete
done
