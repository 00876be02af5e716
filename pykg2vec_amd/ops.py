"""Dispatcher-registered custom ops over the C ABI (BASELINE.json north_star: "exposed to the existing Trainer/Evaluator via a
PyTorch-ROCm custom op that keeps each model's forward()/embed() signature").

  torch.ops.kge.score(key, h, r, t, weights) -> float32 [N]        Model.forward of models/pairwise.py / pointwise.py
  torch.ops.kge.score_backward(key, h, r, t, dscore, weights) -> dense gradients of `weights` (nn.Embedding(sparse=False) semantics)
  torch.ops.kge.one_to_n_scores(x, ent, bias, bf16) -> float32 [B, E]   the projection models' 1-N head (models/projection.py:100-102)
  torch.ops.kge.one_to_n_scores_backward(x, ent, preds, dpreds, need_bias) -> (dx, g_ent, g_bias)

Registered with torch.library.custom_op (schema, fake-tensor kernels, autograd formulas), so the scorer is visible to torch.ops,
torch.library.opcheck and torch.compile (no graph break at `model(h, r, t)`).  The CUDA-key implementations call libkge_hip.so through
pykg2vec_amd.kernels; there is no CPU implementation (a CPU tensor raises, as everywhere in this package).

`key` names the model a call scores with: an integer handle into a weak registry of live model objects (a custom-op argument can only
be a tensor, a number, a string or lists of those; the descriptor needs the model's kernel id, entity / relation counts and
hyper-parameters, which are fixed per model object and therefore trace as a constant)."""
import itertools
import weakref
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import kernels as K

_models = weakref.WeakValueDictionary()
_next_key = itertools.count(1)


def register_model(model):
    """Handle of `model` for torch.ops.kge.score (idempotent; the registry holds the model weakly)."""
    key = getattr(model, "_kge_op_key", None)
    if key is None or _models.get(key) is not model:
        key = next(_next_key)
        _models[key] = model
        model._kge_op_key = key
    return key


def _model(key):
    m = _models.get(int(key))
    if m is None:
        raise RuntimeError("kge::score: model handle %d is not (or no longer) registered" % int(key))
    return m


@torch.library.custom_op("kge::score", mutates_args=(), device_types="cuda")
def score(key: int, h: Tensor, r: Tensor, t: Tensor, weights: List[Tensor]) -> Tensor:
    m = _model(key)
    return K.score_forward(m.make_desc(list(weights)), h.contiguous(), r.contiguous(), t.contiguous())


@score.register_fake
def _(key, h, r, t, weights):
    return h.new_empty((h.numel(),), dtype=torch.float32)


@score.register_kernel("cpu")
def _(key, h, r, t, weights):     # loud, like every other entry of the package: there is no CPU scorer
    raise L.KgeHipError("kge::score: ids and tables must live on the HIP device (got %s); the HIP path has no CPU fallback" % h.device)


@torch.library.custom_op("kge::score_backward", mutates_args=(), device_types="cuda")
def score_backward(key: int, h: Tensor, r: Tensor, t: Tensor, dscore: Tensor, weights: List[Tensor]) -> List[Tensor]:
    m = _model(key)
    grads = [torch.zeros_like(w) for w in weights]
    K.score_backward(m.make_desc(list(weights), grads), h.contiguous(), r.contiguous(), t.contiguous(), dscore.contiguous())
    return grads


@score_backward.register_fake
def _(key, h, r, t, dscore, weights):
    return [torch.empty_like(w) for w in weights]


def _score_setup(ctx, inputs, output):
    key, h, r, t, weights = inputs
    ctx.key = key
    ctx.n_weights = len(weights)
    ctx.save_for_backward(h, r, t, *weights)


def _score_backward(ctx, dscore):
    h, r, t, *weights = ctx.saved_tensors
    return None, None, None, None, score_backward(ctx.key, h, r, t, dscore, weights)


score.register_autograd(_score_backward, setup_context=_score_setup)


@torch.library.custom_op("kge::one_to_n_scores", mutates_args=(), device_types="cuda")
def one_to_n_scores(x: Tensor, ent: Tensor, bias: Optional[Tensor], bf16: bool) -> Tensor:
    b = None if bias is None else bias.contiguous().view(-1)
    return K.head_1n_forward(x.contiguous(), ent.contiguous(), b, precision="bf16" if bf16 else "f32")


@one_to_n_scores.register_fake
def _(x, ent, bias, bf16):
    return x.new_empty((x.shape[0], ent.shape[0]), dtype=torch.float32)


@torch.library.custom_op("kge::one_to_n_scores_backward", mutates_args=(), device_types="cuda")
def one_to_n_scores_backward(x: Tensor, ent: Tensor, preds: Tensor, dpreds: Tensor, need_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    dx, g_ent, g_bias = K.head_1n_backward(x.contiguous(), ent.contiguous(), preds.contiguous(), dpreds.contiguous(), need_bias=need_bias)
    return dx, g_ent, (g_bias if need_bias else ent.new_zeros((ent.shape[0],)))


@one_to_n_scores_backward.register_fake
def _(x, ent, preds, dpreds, need_bias):
    return torch.empty_like(x), torch.empty_like(ent), ent.new_empty((ent.shape[0],))


def _head_setup(ctx, inputs, output):
    x, ent, bias, bf16 = inputs
    ctx.has_bias = bias is not None
    ctx.bias_shape = None if bias is None else bias.shape
    ctx.save_for_backward(x, ent, output)


def _head_backward(ctx, dpreds):
    x, ent, preds = ctx.saved_tensors
    dx, g_ent, g_bias = one_to_n_scores_backward(x, ent, preds, dpreds, ctx.has_bias)
    return dx, g_ent, (g_bias.view(ctx.bias_shape) if ctx.has_bias else None), None


one_to_n_scores.register_autograd(_head_backward, setup_context=_head_setup)
