"""Drop-in pointwise (semantic-matching) models mirroring pykg2vec/models/pointwise.py (DistMult, Complex,
ComplexN3, ANALOGY), scored by HIP kernels.  `get_reg` keeps the reference's tensor-level form for use under the
unmodified reference Trainer; the fused training kernel applies the same regulariser from registers."""
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
