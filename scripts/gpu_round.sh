#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprof kernel trace.  Run via
#   gpurun --timeout 1800 -- 'bash scripts/gpu_round.sh <tag>'
# Everything worth keeping goes to gpurun_out/<tag>/ (merged back into the dev container).
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
{
  echo "== host"; nproc; lscpu | grep -m1 'Model name'; rocminfo | grep -m3 -E 'Marketing Name|gfx' ; rocm-smi --showmeminfo vram 2>/dev/null | tail -3
} > $OUT/host.txt 2>&1
echo "== diag" ; timeout 600 python tests/gpu_diag.py > $OUT/diag.txt 2>&1; echo "diag rc=$?" ; tail -15 $OUT/diag.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.txt
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.txt
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof kernel trace"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
ls $OUT/prof 2>/dev/null | head; find $OUT/prof -name '*kernel_stats*' | head -2 | xargs -r head -30
