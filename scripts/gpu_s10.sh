#!/bin/bash
# the harness's new default (MIOpen immediate mode, naive / GEMM / FFT solver families off) at cfg3, cfg4 and cfg5, stock and fused
TAG=${1:-r7x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
mkdir -p /tmp/miopen/db /tmp/miopen/cache
export MIOPEN_USER_DB_PATH=/tmp/miopen/db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen/cache      # (an EMPTY user db: what a new machine sees)
SECONDS=0
timeout 300 python -m harness.infer --kernel_share > $OUT/infer_stock.json 2> $OUT/infer_stock.err; echo "cfg3 stock rc=$? (${SECONDS}s)"; cut -c1-330 $OUT/infer_stock.json
timeout 300 python -m harness.infer --fused > $OUT/infer_fused.json 2> $OUT/infer_fused.err; echo "cfg3 fused rc=$? (${SECONDS}s)"; cut -c100-330 $OUT/infer_fused.json
timeout 300 python -m harness.train --steps 5 --warmup 2 --kernel_share > $OUT/train_stock.json 2> $OUT/train_stock.err; echo "cfg4 stock rc=$? (${SECONDS}s)"; cut -c100-330 $OUT/train_stock.json
timeout 300 python -m harness.train --steps 5 --warmup 2 --fused --kernel_share > $OUT/train_fused.json 2> $OUT/train_fused.err; echo "cfg4 fused rc=$? (${SECONDS}s)"; cut -c100-330 $OUT/train_fused.json
timeout 300 python -m harness.infer --height 528 --width 960 --batch 2 --fused > $OUT/infer_cfg5_fused.json 2> $OUT/infer_cfg5.err; echo "cfg5 infer fused rc=$? (${SECONDS}s)"; cut -c100-330 $OUT/infer_cfg5_fused.json
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q > $OUT/pytest_model.txt 2>&1; echo "model tests rc=$? (${SECONDS}s)"; tail -1 $OUT/pytest_model.txt
