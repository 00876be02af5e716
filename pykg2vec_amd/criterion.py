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

    @staticmethod
    def multi_class_bce(pred_heads, pred_tails, tr_h, hr_t, label_smoothing, tot_entity):  # criterion.py:41-49
        if label_smoothing is not None and tot_entity is not None:
            hr_t = hr_t * (1.0 - label_smoothing) + 1.0 / tot_entity
            tr_h = tr_h * (1.0 - label_smoothing) + 1.0 / tot_entity
        bce = torch.nn.BCEWithLogitsLoss()  # (sic) applied to sigmoid outputs, as the reference does
        return torch.mean(bce(pred_heads, tr_h)) + torch.mean(bce(pred_tails, hr_t))
