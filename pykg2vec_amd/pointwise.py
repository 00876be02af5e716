"""Drop-in pointwise (semantic-matching) models mirroring pykg2vec/models/pointwise.py (DistMult, Complex,
ComplexN3, ANALOGY, CP, SimplE, SimplE_ignr, QuatE), scored by HIP kernels.  `get_reg` keeps the reference's
tensor-level form for use under the unmodified reference Trainer; the fused training kernel applies the same
regulariser from registers."""
import torch
import torch.nn as nn

from . import _lib as L
from .criterion import Criterion
from .kgmeta import NamedEmbedding, PointwiseModel


def _xavier(*embs):
    for e in embs:
        nn.init.xavier_uniform_(e.weight)


def _power_reg(rows, reg_type, use_abs):
    reg_type = reg_type.lower()
    if reg_type not in ("f2", "n3"):
        raise NotImplementedError("Unknown regularizer type: %s" % reg_type)
    p = 2 if reg_type == "f2" else 3
    total = 0
    for x in rows:
        total = total + torch.sum((x.abs() if use_abs else x) ** p, -1)
    return torch.mean(total)


class DistMult(PointwiseModel):
    """pointwise.py:391-458.  energy = -sum(h * r * t)."""
    kernel_name = "distmult"
    default_reg, reg_abs = "F2", False

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "lmbda"], kwargs))
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, self.hidden_size)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, self.hidden_size)
        _xavier(self.ent_embeddings, self.rel_embeddings)
        self.parameter_list = [self.ent_embeddings, self.rel_embeddings]
        self.loss = Criterion.pointwise_logistic

    def desc_kwargs(self):
        return dict(dim=self.hidden_size)

    def embed(self, h, r, t):
        return self.ent_embeddings(h), self.rel_embeddings(r), self.ent_embeddings(t)

    def reg_rows(self, h, r, t):
        return self.embed(h, r, t)

    def get_reg(self, h, r, t, reg_type=None):
        return self.lmbda * _power_reg(self.reg_rows(h, r, t), reg_type or self.default_reg, self.reg_abs)

    def kernel_lmbda(self):
        """lmbda as the fused kernel applies it: lmbda/n_rows * sum over the gathered rows."""
        return self.lmbda

    def kernel_reg_type(self, reg_type=None):
        rt = (reg_type or self.default_reg).lower()
        if rt == "f2":
            return L.REG_F2
        if rt == "n3":
            return L.REG_N3_ABS if self.reg_abs else L.REG_N3
        raise NotImplementedError("Unknown regularizer type: %s" % rt)


class Complex(DistMult):
    """pointwise.py:122-202.  energy = -Re(<h, r, conj(t)>)."""
    kernel_name = "complex"

    def __init__(self, **kwargs):
        PointwiseModel.__init__(self, self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "lmbda"], kwargs))
        k = self.hidden_size
        self.ent_embeddings_real = NamedEmbedding("emb_e_real", self.tot_entity, k)
        self.ent_embeddings_img = NamedEmbedding("emb_e_img", self.tot_entity, k)
        self.rel_embeddings_real = NamedEmbedding("emb_rel_real", self.tot_relation, k)
        self.rel_embeddings_img = NamedEmbedding("emb_rel_img", self.tot_relation, k)
        _xavier(self.ent_embeddings_real, self.ent_embeddings_img, self.rel_embeddings_real, self.rel_embeddings_img)
        self.parameter_list = [self.ent_embeddings_real, self.ent_embeddings_img, self.rel_embeddings_real,
                               self.rel_embeddings_img]
        self.loss = Criterion.pointwise_logistic

    def embed(self, h, r, t):
        return (self.ent_embeddings_real(h), self.ent_embeddings_img(h), self.rel_embeddings_real(r),
                self.rel_embeddings_img(r), self.ent_embeddings_real(t), self.ent_embeddings_img(t))


class ComplexN3(Complex):
    """pointwise.py:205-238.  Complex with the nuclear 3-norm (|x|^3) regulariser by default."""
    default_reg, reg_abs = "N3", True

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.model_name = "complexn3"


class ANALOGY(DistMult):
    """pointwise.py:13-119.  DistMult term on k-dim tables + ComplEx term on k/2-dim tables."""
    kernel_name = "analogy"

    def __init__(self, **kwargs):
        PointwiseModel.__init__(self, self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "lmbda"], kwargs))
        k = self.hidden_size
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, k)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, k)
        self.ent_embeddings_real = NamedEmbedding("emb_e_real", self.tot_entity, k // 2)
        self.ent_embeddings_img = NamedEmbedding("emb_e_img", self.tot_entity, k // 2)
        self.rel_embeddings_real = NamedEmbedding("emb_rel_real", self.tot_relation, k // 2)
        self.rel_embeddings_img = NamedEmbedding("emb_rel_img", self.tot_relation, k // 2)
        self.parameter_list = [self.ent_embeddings, self.rel_embeddings, self.ent_embeddings_real,
                               self.ent_embeddings_img, self.rel_embeddings_real, self.rel_embeddings_img]
        _xavier(*self.parameter_list)
        self.loss = Criterion.pointwise_logistic

    def embed_complex(self, h, r, t):
        return (self.ent_embeddings_real(h), self.ent_embeddings_img(h), self.rel_embeddings_real(r),
                self.rel_embeddings_img(r), self.ent_embeddings_real(t), self.ent_embeddings_img(t))

    def reg_rows(self, h, r, t):
        return tuple(self.embed_complex(h, r, t)) + tuple(self.embed(h, r, t))


class CP(DistMult):
    """pointwise.py:320-387.  Canonical tensor decomposition: separate subject / object entity tables."""
    kernel_name = "cp"
    default_reg, reg_abs = "N3", False

    def __init__(self, **kwargs):
        PointwiseModel.__init__(self, self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "lmbda"], kwargs))
        k = self.hidden_size
        self.sub_embeddings = NamedEmbedding("sub_embedding", self.tot_entity, k)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, k)
        self.obj_embeddings = NamedEmbedding("obj_embedding", self.tot_entity, k)
        _xavier(self.sub_embeddings, self.rel_embeddings, self.obj_embeddings)
        self.parameter_list = [self.sub_embeddings, self.rel_embeddings, self.obj_embeddings]
        self.loss = Criterion.pointwise_logistic

    def embed(self, h, r, t):
        return self.sub_embeddings(h), self.rel_embeddings(r), self.obj_embeddings(t)


class SimplE(DistMult):
    """pointwise.py:461-546.  energy = -clamp(<head[h], rel[r], tail[t]> + <head[t], rel_inv[r], tail[h]> / 2, +-20)."""
    kernel_name = "simple"

    def __init__(self, **kwargs):
        PointwiseModel.__init__(self, self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "lmbda"], kwargs))
        k = self.hidden_size
        self.tot_train_triples = kwargs["tot_train_triples"]
        self.batch_size = kwargs["batch_size"]
        self.ent_head_embeddings = NamedEmbedding("ent_head_embedding", self.tot_entity, k)
        self.ent_tail_embeddings = NamedEmbedding("ent_tail_embedding", self.tot_entity, k)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, k)
        self.rel_inv_embeddings = NamedEmbedding("rel_inv_embedding", self.tot_relation, k)
        self.parameter_list = [self.ent_head_embeddings, self.ent_tail_embeddings, self.rel_embeddings,
                               self.rel_inv_embeddings]
        _xavier(*self.parameter_list)
        self.loss = Criterion.pointwise_logistic

    def embed(self, h, r, t):
        return (self.ent_head_embeddings(h), self.ent_head_embeddings(t), self.rel_embeddings(r),
                self.rel_inv_embeddings(r), self.ent_tail_embeddings(t), self.ent_tail_embeddings(h))

    def get_reg(self, h, r, t, reg_type="F2"):
        """pointwise.py:528-536: the reference hands the ID tensors to get_reg and regularises THEM -- a constant
        lmbda * (sum h^p + sum r^p + sum t^p) in float32 with no gradient.  Kept, so losses match."""
        rt = reg_type.lower()
        if rt not in ("f2", "n3"):
            raise NotImplementedError("Unknown regularizer type: %s" % reg_type)
        p = 2 if rt == "f2" else 3
        term = torch.sum(h.float() ** p, -1) + torch.sum(r.float() ** p, -1) + torch.sum(t.float() ** p, -1)
        return self.lmbda * torch.mean(term)

    def kernel_reg_type(self, reg_type="F2"):
        return L.REG_ID_F2 if reg_type.lower() == "f2" else L.REG_ID_N3  # loss constant only; no parameter gradient


class SimplE_ignr(SimplE):
    """pointwise.py:549-592.  Same tables; both terms summed without the 1/2."""
    kernel_name = "simple_ignr"

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.model_name = "simple_ignr"

    def embed(self, h, r, t):
        cat = lambda e1, i1, e2, i2: torch.cat([e1.weight.index_select(0, i1), e2.weight.index_select(0, i2)], 1)
        return (cat(self.ent_head_embeddings, h, self.ent_head_embeddings, t),
                cat(self.rel_embeddings, r, self.rel_inv_embeddings, r),
                cat(self.ent_tail_embeddings, t, self.ent_tail_embeddings, h))


class QuatE(DistMult):
    """pointwise.py:595-768.  energy = -sum((h (x) r/|r|) . t) with element-wise quaternions (s, x, y, z)."""
    kernel_name = "quate"
    default_reg, reg_abs = "N3", True

    def __init__(self, **kwargs):
        PointwiseModel.__init__(self, self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "lmbda"], kwargs))
        k = self.hidden_size
        for c in "sxyz":
            setattr(self, "ent_%s_embedding" % c, NamedEmbedding("ent_%s_embedding" % c, self.tot_entity, k))
        # the reference overwrites the four rel_{s,x,y,z} weights with _quaternion_init(tot_entity, k) arrays before
        # the Xavier pass (pointwise.py:653-668), so those tables (and its checkpoints) carry tot_entity rows
        for c in "sxyz":
            setattr(self, "rel_%s_embedding" % c, NamedEmbedding("rel_%s_embedding" % c, self.tot_entity, k))
        self.rel_w_embedding = NamedEmbedding("rel_w_embedding", self.tot_relation, k)
        self.fc = nn.Linear(100, 50, bias=False)  # unused by forward; present in the reference's state_dict
        self.ent_dropout = nn.Dropout(0)
        self.rel_dropout = nn.Dropout(0)
        self.bn = nn.BatchNorm1d(k)
        self.parameter_list = [getattr(self, "%s_%s_embedding" % (a, c)) for a in ("ent", "rel") for c in "sxyz"]
        self.parameter_list.append(self.rel_w_embedding)
        _xavier(*self.parameter_list)
        self.loss = Criterion.pointwise_logistic

    def embed(self, h, r, t):
        e = [getattr(self, "ent_%s_embedding" % c) for c in "sxyz"]
        q = [getattr(self, "rel_%s_embedding" % c) for c in "sxyz"]
        return tuple(x(h) for x in e) + tuple(x(t) for x in e) + tuple(x(r) for x in q)

    def get_reg(self, h, r, t, reg_type=None):
        rt = (reg_type or self.default_reg).lower()
        if rt not in ("f2", "n3"):
            raise NotImplementedError("Unknown regularizer type: %s" % rt)
        p = 2 if rt == "f2" else 3
        return self.lmbda * sum(torch.mean(torch.abs(x) ** p) for x in self.embed(h, r, t))

    def kernel_lmbda(self):
        return self.lmbda / self.hidden_size  # means over all B*k elements, not over the B rows

    def kernel_reg_type(self, reg_type=None):
        rt = (reg_type or self.default_reg).lower()
        if rt == "f2":
            return L.REG_F2
        if rt == "n3":
            return L.REG_N3_ABS
        raise NotImplementedError("Unknown regularizer type: %s" % rt)
