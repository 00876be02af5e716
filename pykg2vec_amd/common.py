"""Enums shared with the reference (pykg2vec/common.py:12-24).

When the reference package is importable its own enum classes are re-used, so that the unmodified reference
Trainer / Generator (`model.training_strategy == TrainingStrategy.PAIRWISE_BASED`, utils/trainer.py:274-296,
data/generator.py:301-309) recognise our drop-in models; otherwise identical stand-ins are defined.
"""
from enum import Enum

try:  # integration mode: living next to the reference
    from pykg2vec.common import Monitor, TrainingStrategy  # type: ignore
except Exception:  # standalone (e.g. the GPU box)
    class Monitor(Enum):
        MEAN_RANK = "mr"
        FILTERED_MEAN_RANK = "fmr"
        MEAN_RECIPROCAL_RANK = "mrr"
        FILTERED_MEAN_RECIPROCAL_RANK = "fmrr"

    class TrainingStrategy(Enum):
        PROJECTION_BASED = "projection_based"
        PAIRWISE_BASED = "pairwise_based"
        POINTWISE_BASED = "pointwise_based"
