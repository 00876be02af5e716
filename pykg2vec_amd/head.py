"""The 1-N scoring head the projection models end their forward with (pykg2vec/models/projection.py:100-102,
335-336, 444-447, 606-609, 734-737): `sigmoid(x @ ent_embeddings.weight.T + b)`, on the MI355X matrix cores.

`one_to_n_scores` is a differentiable drop-in for those three lines (x comes from the model's own torch layers; the
entity table and the bias receive dense gradients like nn.Embedding(sparse=False)); `multi_class_bce_step` is the fused
training form: head + one direction of Criterion.multi_class_bce + backward without a [B, E] prediction tensor."""
import torch

from . import kernels as K
from . import ops  # noqa: F401  (registers torch.ops.kge.*)


def one_to_n_scores(x, ent_weight, bias=None, precision="f32"):
    """sigmoid(x @ ent_weight.T + bias) -> [B, E]; bias: [E] or [1, E] (ConvE's `b.weight`) or None (TuckER).
    precision="bf16": the forward GEMM with bfloat16-rounded operands and fp32 accumulation (v_mfma_f32_32x32x16_bf16; the [B, E]
    output write then bounds it instead of the f32 matrix cores) -- for scoring all entities at evaluation time or where the
    model tolerates it; the backward GEMMs stay fp32 and use the saved predictions.  Default "f32" = the reference's arithmetic."""
    if precision not in ("f32", "bf16"):
        raise ValueError("precision must be 'f32' or 'bf16'")
    return torch.ops.kge.one_to_n_scores(x, ent_weight, bias, precision == "bf16")


def multi_class_bce_step(x, ent_weight, bias, label_off, label_ids, label_smoothing, loss_buf, g_ent, g_bias=None):
    """Fused head + one direction of Criterion.multi_class_bce (utils/criterion.py:41-49) + backward.
    Labels are the batch's hr_t (tail direction) or tr_h (head direction) rows as CSR.  Returns d loss / d x."""
    b = None if bias is None else bias.contiguous().view(-1)
    gb = None if g_bias is None else g_bias.view(-1)
    return K.head_1n_bce(x.contiguous(), ent_weight.contiguous(), b, label_off, label_ids, label_smoothing, loss_buf, g_ent, gb)


def one_to_n_rank(x, ent_weight, bias, truth, known_off=None, known_ids=None, return_ties=False):
    """Evaluation form of the head (models/projection.py:119-125 + utils/evaluator.py:70-123): int32 [2, B] = for every row the number
    of entities predicted strictly above the true one, and the same count without the row's known entities (CSR known_off int64
    [B+1] / known_ids int32).  No [B, E] tensor is formed: the rank sweep's tiles count against the true entity's prediction."""
    b = None if bias is None else bias.contiguous().view(-1)
    return K.head_1n_rank(x.contiguous(), ent_weight.contiguous(), b, truth.contiguous(), known_off, known_ids, return_ties=return_ties)
