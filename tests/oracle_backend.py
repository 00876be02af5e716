"""TEST-ONLY compute backend: the numpy oracle behind the interface `pykg2vec_amd.trainer.Trainer` expects from
`pykg2vec_amd.kernels`.  It lets the multi-process plumbing (batch sharding, gradient all-reduce, replica
consistency) run under gloo on CPU.  Never imported by the product."""
import numpy as np
import torch

import kge_oracle as ko

OPTIMIZER_IDS = {"sgd": 0, "adam": 1, "adagrad": 2, "rms": 3}


class Desc:
    def __init__(self, model, weights, grads):
        self.name = model.kernel_name if model.model_name != "complexn3" else "complexn3"
        self.names = [n.split(".")[0] for n, _ in model.named_parameters()]
        self.weights, self.grads = weights, grads
        self.hp = {k: getattr(model, k) for k in ("l1_flag", "margin", "hidden_size", "lmbda") if hasattr(model, k)}

    def params(self):
        return {n: w.detach().numpy() for n, w in zip(self.names, self.weights)}


def model_desc(model, weights=None, grads=None):
    if weights is None:
        weights = [p.weight.data for p in model.parameter_list]
    return Desc(model, list(weights), None if grads is None else list(grads))


def new_loss_buffer(device):
    return torch.zeros(1024, dtype=torch.float32)


def read_loss(buf):
    return buf.view(32, 32)[:, 0].sum()


def _np(t):
    return t.numpy()


def _add_grads(desc, G):
    for n, g in zip(desc.names, desc.grads):
        g += torch.from_numpy(np.ascontiguousarray(G[n]))


def train_pairwise_hinge(desc, ph, pr, pt, nh, nr, nt, margin, loss_buf):
    if ph.numel() == 0:
        return
    hp = dict(desc.hp, margin=margin)
    loss, G, _, _ = ko.train_step_grads(desc.name, desc.params(), tuple(map(_np, (ph, pr, pt, nh, nr, nt))), **hp)
    _add_grads(desc, G)
    loss_buf[0] += float(loss)


def train_pairwise_selfadv(desc, ph, pr, pt, nh, nr, nt, neg_rate, alpha, loss_buf, workspace=None):
    hp = dict(desc.hp, neg_rate=neg_rate, alpha=alpha)
    loss, G, _, _ = ko.train_step_grads(desc.name, desc.params(), tuple(map(_np, (ph, pr, pt, nh, nr, nt))), **hp)
    _add_grads(desc, G)
    loss_buf[0] += float(loss)
    return workspace


def train_pointwise_logistic(desc, h, r, t, y, lmbda, reg_type, loss_buf, bundle=1):
    hp = dict(desc.hp, lmbda=lmbda)
    loss, G, _, _ = ko.train_step_grads(desc.name, desc.params(), tuple(map(_np, (h, r, t, y))), **hp)
    _add_grads(desc, G)
    loss_buf[0] += float(loss)


def optimizer_step(kind, param, grad, state1, state2, lr, step, zero_grad=True, dev_hyper=None):
    P = {"p": param.numpy()}
    st = {"step": step - 1}
    if kind == "adam":
        st["m"], st["v"] = {"p": state1.numpy()}, {"p": state2.numpy()}
    elif kind in ("adagrad", "rms"):
        st["sq"] = {"p": state1.numpy()}
    ko.optimizer_step(kind, P, {"p": grad.numpy()}, st, lr)
    if zero_grad:
        grad.zero_()


def rescal_normalize(ent, rel, k):
    for w in (ent, rel):
        w /= w.norm(dim=-1, keepdim=True)


def triple_set_build(triples):
    return {tuple(map(int, x)) for x in triples.numpy()}


def sample_batch(triples, perm, start, n_pos, neg_rate, tot_entity, bern_prob, slots, seed, offset, pointwise=False,
                 out=None, cursor=None):
    """Deterministic function of (seed, offset + slot index) like the device sampler (different stream)."""
    pos = triples[perm[start:start + n_pos]].numpy()
    nh, nr, nt = [], [], []
    for i in range(n_pos):
        h, r, t = map(int, pos[i])
        for k in range(neg_rate):
            rng = np.random.default_rng([int(seed), int(offset) + i * neg_rate + k])
            tail = rng.random() > (0.5 if bern_prob is None else float(bern_prob[r]))
            while True:
                e = int(rng.integers(tot_entity))
                c = (h, r, e) if tail else (e, r, t)
                if c not in slots:
                    break
            nh.append(c[0]); nr.append(c[1]); nt.append(c[2])
    nh, nr, nt = (np.asarray(a, dtype=np.int64) for a in (nh, nr, nt))
    if pointwise:
        return [torch.from_numpy(a) for a in ko.pointwise_layout(pos, nh, nr, nt, neg_rate)]
    return [torch.from_numpy(np.ascontiguousarray(a)) for a in (pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt)]


def eval_ranks(desc, triples, tail_off, tail_ids, head_off, head_ids, workspace=None):
    trip = triples.numpy()
    hr_t, tr_h = {}, {}
    to, ti, ho, hi = (x.numpy() for x in (tail_off, tail_ids, head_off, head_ids))
    for i, (h, r, t) in enumerate(trip):
        hr_t[(int(h), int(r))] = set(map(int, ti[to[i]:to[i + 1]]))
        tr_h[(int(t), int(r))] = set(map(int, hi[ho[i]:ho[i + 1]]))
    _, rk = ko.evaluate(desc.name, desc.params(), trip, hr_t, tr_h, **desc.hp)
    return torch.from_numpy(np.stack([rk["head"], rk["tail"], rk["fhead"], rk["ftail"]]).astype(np.int32))
