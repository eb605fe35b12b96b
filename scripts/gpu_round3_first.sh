#!/bin/bash
# First GPU call of the next round: everything that was built at the end of round 2 without a GPU budget to run it.
#   gpurun --timeout 900 -- 'bash scripts/gpu_round3_first.sh r3a'
# 1. pytest -m gpu (ops) -- the state the round starts from
# 2. scripts/check_lga_paired.py   GANET_LGA_PAIRED=1: Lga2Function parity on vs off + timing (kernels lga_apply_pp_pi/_po, lga_filter_grad_pp_xp/_gyp)
# 3. bench.py with GANET_LGA_PAIRED=0 and =1 (same box A/B of the headline number)
# 3b. bench.py with GANET_LGA_MIX=1 (mixed item list), alone and with PAIRED; LGA parity tests under GANET_LGA_MIX=1
# 4. scripts/check_wide_col.py     GANET_SGA_WIDE_COL=1: parity + timing of the vertical scans on the stress shape
TAG=${1:-r3a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu (ops)"
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_model.py > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -3 $OUT/pytest_gpu.txt
echo "== LGA2 with a pair-interleaved intermediate"
timeout 600 python scripts/check_lga_paired.py > $OUT/check_lga_paired.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/check_lga_paired.txt | tail -24
echo "== bench, GANET_LGA_PAIRED=0 / 1"
GANET_LGA_PAIRED=0 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_paired0.json 2> $OUT/bench_paired0.err; echo "rc=$?"
GANET_LGA_PAIRED=1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_paired1.json 2> $OUT/bench_paired1.err; echo "rc=$?"
python - <<PY
import json
for k in (0, 1):
    try:
        d = json.load(open("$OUT/bench_paired%d.json" % k))
        print("GANET_LGA_PAIRED=%d: %.1f cv/s  %.4f ms per step (through the autograd Functions; stage_ms times the API-layout entries either way)" % (k, d["value"], d["ms_per_step"]))
    except Exception as e:
        print("GANET_LGA_PAIRED=%d: no result (%r)" % (k, e))
PY
echo "== bench, GANET_LGA_MIX=1 (mixed item list of the plane-pair forward / data-backward), and MIX + PAIRED"
GANET_LGA_MIX=1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_mix1.json 2> $OUT/bench_mix1.err; echo "rc=$?"
GANET_LGA_MIX=1 GANET_LGA_PAIRED=1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_mix1_paired1.json 2> $OUT/bench_mix1_paired1.err; echo "rc=$?"
python - <<PY
import json
for k in ("mix1", "mix1_paired1"):
    try:
        d = json.load(open("$OUT/bench_%s.json" % k))
        print("%s: %.1f cv/s  %.4f ms per step; stage lga fwd %.4f bwd %.4f" % (k, d["value"], d["ms_per_step"], d["stage_ms"]["lga_fwd_pass"], d["stage_ms"]["lga_bwd_pass"]))
    except Exception as e:
        print("%s: no result (%r)" % (k, e))
PY
GANET_LGA_MIX=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lga" > $OUT/pytest_lga_mix.txt 2>&1; echo "lga parity with GANET_LGA_MIX=1 rc=$?"; tail -2 $OUT/pytest_lga_mix.txt
echo "== wide column blocks"
timeout 600 python scripts/check_wide_col.py > $OUT/check_wide_col.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/check_wide_col.txt | tail -20
echo "== done"
