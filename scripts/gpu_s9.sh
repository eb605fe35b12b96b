#!/bin/bash
TAG=${1:-r7t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_pixels" > $OUT/pytest_p2.txt 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_p2.txt
timeout 600 python scripts/ab_step.py libganet_hip_new.so libganet_hip_new.so libganet_hip_new.so@GANET_SGA_POINT2=1 libganet_hip_p2w3.so@GANET_SGA_POINT2=1 libganet_hip_p2w5.so@GANET_SGA_POINT2=1 libganet_hip_new.so libganet_hip_new.so@GANET_SGA_POINT2=1 libganet_hip_p2w3.so@GANET_SGA_POINT2=1 > $OUT/ab_step.txt 2>&1; echo rc=$?; tail -8 $OUT/ab_step.txt
timeout 300 python scripts/ab_sga_stages.py libganet_hip_new.so libganet_hip_new.so@GANET_SGA_POINT2=1 libganet_hip_p2w3.so@GANET_SGA_POINT2=1 > $OUT/ab_sga_stages.txt 2>&1; grep -o "^[^ ]* \|'sga_bwd_point': [0-9.]*" $OUT/ab_sga_stages.txt | paste - - | tail -6
