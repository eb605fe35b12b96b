#!/bin/bash
# Same-box A/B of library builds (GPU boxes differ by +-8 %): bash scripts/ab_libs.sh <prefix> libA.so libB.so ...
# runs scripts/bench_stages.py <prefix> for each library, twice, interleaved.
PRE=$1; shift
cp ganet_amd/libganet_hip.so /tmp/_keep.so
for rep in 1 2; do
  for L in "$@"; do
    cp $L ganet_amd/libganet_hip.so
    echo "$(basename $L) #$rep: $(python scripts/bench_stages.py $PRE 2>&1 | tail -1)"
  done
done
cp /tmp/_keep.so ganet_amd/libganet_hip.so
