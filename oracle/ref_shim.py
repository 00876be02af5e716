"""Import shim for the live reference (TEST INFRASTRUCTURE ONLY, container-only).

`/root/reference` is Sujit-O/pykg2vec, pure Python on PyTorch.  Two of its
module-level imports are absent offline (`hyperopt`: pykg2vec/common.py:8-9,
`seaborn`: pykg2vec/utils/visualization.py:7); neither is touched on the
scoring / loss / ranking path, so inert stand-ins are installed before import.

Only `oracle/make_golden.py` and container-local cross-checks in `tests/`
use this file; nothing that runs on the GPU box may (the reference does not
exist there).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PYKG2VEC_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pykg2vec"))


class _Inert:
    def __getattr__(self, _name):
        return lambda *a, **k: None


def install():
    """Make `import pykg2vec` resolve to the read-only reference tree."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if "hyperopt" not in sys.modules:
        hy = types.ModuleType("hyperopt")
        hy.hp = _Inert()
        for n in ("fmin", "tpe", "Trials", "STATUS_OK", "space_eval"):
            setattr(hy, n, None)
        pyll = types.ModuleType("hyperopt.pyll")
        base = types.ModuleType("hyperopt.pyll.base")
        base.scope = _Inert()
        sys.modules.update({"hyperopt": hy, "hyperopt.pyll": pyll, "hyperopt.pyll.base": base})
    if "seaborn" not in sys.modules:
        sb = types.ModuleType("seaborn")
        sb.set_style = lambda *a, **k: None
        sys.modules["seaborn"] = sb
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
