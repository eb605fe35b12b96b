"""Locates the reference's model definitions and imports them ON TOP of this repository's `libs/` drop-in.

The model files are the reference's (never vendored).  Search order for the directory that contains `models/`:
  1. $GANET_REF_ROOT                       a GANet checkout of the user's
  2. /root/reference                       the build container
  3. <repo>/oracle/_ref/pyref              byte-compiled copies made by `make -C oracle` where (2) exists -- a git-ignored
                                           build product that travels to the GPU box (oracle/compile_pyref.py)
`libs.GANet.modules.GANet` / `libs.sync_bn.modules.sync_bn` always resolve to THIS repository's libs/ (placed first on
sys.path), whatever the model root also contains -- that is INTEGRATION.md's mode A.
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = {"GANet_deep": "models.GANet_deep", "GANet11": "models.GANet11"}


def model_root():
    cands = [os.environ.get("GANET_REF_ROOT"), "/root/reference", os.path.join(ROOT, "oracle", "_ref", "pyref")]
    for c in cands:
        if c and any(os.path.exists(os.path.join(c, "models", "GANet_deep" + ext)) for ext in (".py", ".pyc")):
            return c
    return None


def available():
    return model_root() is not None


def model_class(name="GANet_deep"):
    """The reference's `GANet` class of models/<name>.py, importing the drop-in ops through `libs.*`."""
    if name not in MODELS:
        raise ValueError(f"unknown model {name!r}; the reference ships {sorted(MODELS)}")
    root = model_root()
    if root is None:
        raise RuntimeError("no GANet model code found: set GANET_REF_ROOT to a feihuzhang/GANet checkout "
                           "(its models/ directory is the caller of these ops; it is not part of this repository)")
    if ROOT in sys.path:
        sys.path.remove(ROOT)
    sys.path.insert(0, ROOT)                       # this repo's libs/ first
    if root not in sys.path:
        sys.path.append(root)
    import libs.GANet.modules.GANet as drop_in     # noqa: F401  (fails loudly if the drop-in is not importable)
    if not os.path.realpath(drop_in.__file__).startswith(os.path.realpath(ROOT)):
        raise RuntimeError(f"`libs` resolved to {drop_in.__file__}, not to this repository's drop-in")
    mod = importlib.import_module(MODELS[name])
    return mod.GANet
