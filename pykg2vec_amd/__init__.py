"""pykg2vec_amd -- MI355X-native scoring / training / ranking path behind pykg2vec's model API.

The package is the host-side mirror of the reference interface for this path only; all arithmetic is in
libkge_hip.so (C ABI: include/kge_hip.h).  Importing the package does not need a GPU; calling it does.
"""
from . import _lib  # noqa: F401
from .common import Monitor, TrainingStrategy  # noqa: F401

__all__ = ["pairwise", "pointwise", "criterion", "evaluator", "trainer", "generator", "kernels", "head"]

MODEL_MAP = {  # lower-case name -> "module.Class", same keys as Importer.modelMap (pykg2vec/common.py:266-298) for this path
    "transe": "pairwise.TransE", "transh": "pairwise.TransH", "transd": "pairwise.TransD", "rotate": "pairwise.RotatE",
    "rescal": "pairwise.Rescal", "ntn": "pairwise.NTN", "distmult": "pointwise.DistMult", "complex": "pointwise.Complex",
    "complexn3": "pointwise.ComplexN3", "analogy": "pointwise.ANALOGY",
    "transm": "pairwise.TransM", "transr": "pairwise.TransR", "cp": "pointwise.CP", "simple": "pointwise.SimplE",
    "simple_ignr": "pointwise.SimplE_ignr", "quate": "pointwise.QuatE",
}


def import_model(name):
    """Importer().import_model_config(name) analogue (pykg2vec/common.py:300-328) returning the model class."""
    import importlib
    try:
        mod, cls = MODEL_MAP[name.lower()].split(".")
    except KeyError:
        raise ValueError("%s model has not been implemented on the MI355X path." % name)
    return getattr(importlib.import_module("pykg2vec_amd." + mod), cls)
