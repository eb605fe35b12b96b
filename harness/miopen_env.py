"""MIOpen set-up for the full-model harness.

The 3-D convolutions of the cost aggregation dominate a GANet pass once the GA ops are fast, and this
image ships no gfx950 find-db / kernel-db: in PyTorch's default "immediate" mode MIOpen picks solvers from a heuristic that
is badly off for these shapes (measured on MI355X, profiles/r2_model_*.json: GANet-deep training step 1,838 ms immediate vs
113 ms with torch.backends.cudnn.benchmark = True; inference 104.6 vs 58.2 ms).  Find mode times the applicable solvers once
per convolution shape (minutes for the ~200 shapes of a training step) and records the winners in MIOpen's USER db, so the
harness keeps that db inside the checkout (<repo>/miopen_cache, git-ignored) unless the caller already points
MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR somewhere: the search is paid once per machine, not per run.

Round 4: the search is no longer the default.  What made immediate mode slow is ONE thing: with no find-db, MIOpen's fallback
takes the first applicable solver, and for these 3-D convolutions that is the naive direct kernel (or im2col + GEMM).  With those
families (and FFT) switched off through MIOpen's own per-family switches, the first applicable solver is a CK implicit-GEMM,
Winograd or direct-asm kernel -- one compilation per problem, no benchmarking -- and the step is as fast as after a full search
(cfg4, fused call sites: 87.3 ms immediate vs 87.0 ms find mode, profiles/r7w_*, r7_model_train_fused.json), while the cfg5
training step, whose find pass outlasted 200 s, 1,000 s and 1,680 s in three rounds, completes its first step in 14 s.
`disable_slow_solver_families()` sets those switches (setdefault: the caller's environment wins); `--miopen_find` still runs the
search (with the same families off it is shorter, but still minutes for a new set of training shapes)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_repo_miopen_cache():
    base = os.path.join(ROOT, "miopen_cache")
    for var, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
        if var not in os.environ:
            path = os.path.join(base, sub)
            try:
                os.makedirs(path, exist_ok=True)
            except OSError:
                continue
            os.environ[var] = path


SLOW_FAMILIES = ("MIOPEN_DEBUG_CONV_GEMM", "MIOPEN_DEBUG_CONV_FFT", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD",
                 "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW")


def disable_slow_solver_families():
    """Before torch loads MIOpen: take the naive direct, GEMM and FFT convolution solvers out of MIOpen's lists (its documented
    per-family debug switches), so that immediate mode's "first applicable solver" is a fast one.  GANET_MIOPEN_ALL_SOLVERS=1
    leaves MIOpen as it ships."""
    if os.environ.get("GANET_MIOPEN_ALL_SOLVERS", "0") == "1":
        return
    for var in SLOW_FAMILIES:
        os.environ.setdefault(var, "0")

