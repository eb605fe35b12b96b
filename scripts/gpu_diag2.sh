#!/bin/bash
TAG=${1:-r7n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 300 python scripts/diag_fg.py libganet_hip_base5.so libganet_hip_new.so > $OUT/diag_fg.txt 2>&1; echo rc=$?; grep -v amdgpu.ids $OUT/diag_fg.txt | cut -c1-300
( cd /tmp; for L in base5 new; do timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_$L -o t --output-format csv -- python $ROOT/scripts/bench_lib.py libganet_hip_$L.so --no-cpu-baseline --no-roofline --no-overlap > $OUT/prof_$L.log 2>&1; done )
for L in base5 new; do echo $L; grep "lga_filter_grad" $OUT/prof_$L/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-200; done
find $OUT -name '*kernel_trace.csv' -delete
