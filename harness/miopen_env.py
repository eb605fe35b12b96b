"""MIOpen set-up for the full-model harness.

The 3-D convolutions of the cost aggregation dominate a GANet pass once the GA ops are fast, and this
image ships no gfx950 find-db / kernel-db: in PyTorch's default "immediate" mode MIOpen picks solvers from a heuristic that
is badly off for these shapes (measured on MI355X, profiles/r2_model_*.json: GANet-deep training step 1,838 ms immediate vs
113 ms with torch.backends.cudnn.benchmark = True; inference 104.6 vs 58.2 ms).  Find mode times the applicable solvers once
per convolution shape (minutes for the ~200 shapes of a training step) and records the winners in MIOpen's USER db, so the
harness turns it on by default and keeps that db inside the checkout (<repo>/miopen_cache, git-ignored) unless the caller
already points MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR somewhere: the search is paid once per machine, not per run."""
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
