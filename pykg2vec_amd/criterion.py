"""Loss functions with the reference's signatures (pykg2vec/utils/criterion.py:13-34).

These tensor-level forms exist so that the UNMODIFIED reference Trainer (`loss = self.model.loss(pos, neg, margin)`,
utils/trainer.py:147-180) can drive the drop-in models; they act on [B] score vectors only.  The MI355X training
path (pykg2vec_amd.trainer.Trainer) does not call them: there the loss is fused into the scoring kernel
(kge_train_pairwise_hinge / _selfadv / kge_train_pointwise_logistic in include/kge_hip.h).
"""
import torch
import torch.nn.functional as F


class Criterion:
    @staticmethod
    def pariwise_logistic(pos_preds, neg_preds, neg_rate, alpha):  # (sic) reference spelling, criterion.py:13
        neg = (-neg_preds).view(-1, neg_rate)
        weights = torch.softmax(neg * alpha, dim=1).detach()
        neg_term = (weights * F.logsigmoid(-neg)).sum(dim=-1)
        return -neg_term.mean() - F.logsigmoid(-pos_preds).mean()

    @staticmethod
    def pairwise_hinge(pos_preds, neg_preds, margin):  # criterion.py:25-29
        return torch.clamp_min(pos_preds + margin - neg_preds, 0).sum()

    @staticmethod
    def pointwise_logistic(preds, target):  # criterion.py:31-34
        return F.softplus(target * preds).mean()
