"""TEST / HARNESS INFRASTRUCTURE.  Byte-compiles the reference's Python layer from where it lies (argv[1], normally
/root/reference) into sourceless .pyc modules under argv[2] (oracle/_ref/pyref, git-ignored like the rest of oracle/_ref):
the reference checkout does not exist on the GPU box, its bytecode -- a build product, like _ref/libganet_ref.so -- travels.
No reference source text is written anywhere.

  <out>/models/{GANet_deep,GANet11}.pyc              the callers of the ops (models/GANet_deep.py, models/GANet11.py)
  <out>/modeB/libs/GANet/functions/GANet.pyc         the reference's OWN autograd Functions (libs/GANet/functions/GANet.py),
  <out>/modeB/libs/GANet/modules/GANet.pyc           and modules, for running them unmodified on top of this repo's pybind
                                                     module `GANet` (INTEGRATION.md mode B)
"""
import os
import py_compile
import sys

PLAN = [
    ("models/GANet_deep.py", "models/GANet_deep.pyc"),
    ("models/GANet11.py", "models/GANet11.pyc"),
    ("libs/GANet/functions/GANet.py", "modeB/libs/GANet/functions/GANet.pyc"),
    ("libs/GANet/modules/GANet.py", "modeB/libs/GANet/modules/GANet.pyc"),
]


def main(ref, out):
    for src, dst in PLAN:
        dst = os.path.join(out, dst)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: what tracebacks show -- the path inside the reference checkout
        py_compile.compile(os.path.join(ref, src), cfile=dst, dfile=os.path.join("<reference>", src), doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    # package markers (empty modules; nothing of the reference in them)
    for pkg in ("models", "modeB/libs", "modeB/libs/GANet", "modeB/libs/GANet/functions", "modeB/libs/GANet/modules"):
        open(os.path.join(out, pkg, "__init__.py"), "w").close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
