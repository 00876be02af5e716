"""Drop-in pairwise (translational-distance) models: same class names, constructor kwargs, parameter names,
`forward/embed/loss/get_reg/parameter_list` contract as pykg2vec/models/pairwise.py, scored by HIP kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from .criterion import Criterion
from .kgmeta import NamedEmbedding, PairwiseModel

_PI = 3.14159265358979323846


def _xavier(*embs):
    for e in embs:
        nn.init.xavier_uniform_(e.weight)


class TransE(PairwiseModel):
    """pairwise.py:12-93.  energy = || h^ + r^ - t^ ||_{1|2} with x^ = F.normalize(x)."""
    kernel_name = "transe"

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "l1_flag"], kwargs))
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, self.hidden_size)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, self.hidden_size)
        _xavier(self.ent_embeddings, self.rel_embeddings)
        self.parameter_list = [self.ent_embeddings, self.rel_embeddings]
        self.loss = Criterion.pairwise_hinge

    def desc_kwargs(self):
        return dict(dim=self.hidden_size, l1_flag=bool(self.l1_flag))

    def embed(self, h, r, t):
        return self.ent_embeddings(h), self.rel_embeddings(r), self.ent_embeddings(t)


class TransM(TransE):
    """pairwise.py:281-365.  TransE distance weighted by a fixed per-relation theta_r computed from the train split."""
    kernel_name = "transm"

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        train = kwargs["knowledge_graph"].read_cache_data("triplets_train")
        self.theta = torch.from_numpy(self._theta(train, self.tot_relation)).to(kwargs["device"])

    @staticmethod
    def _theta(train, tot_relation):
        """pairwise.py:303-315.  rel_head / rel_tail there are per-triple LISTS, so both lengths equal the relation's
        triple count c:  theta_r = 1 / log(2 + c/(1+c) + c/(1+c))  (float64, stored as float32)."""
        import numpy as np
        if len(train) and hasattr(train[0], "r"):
            rel = np.fromiter((t.r for t in train), dtype=np.int64, count=len(train))
        else:
            rel = np.asarray(train, dtype=np.int64).reshape(-1, 3)[:, 1]
        c = np.bincount(rel, minlength=tot_relation).astype(np.float64)
        return (1.0 / np.log(2.0 + c / (1.0 + c) + c / (1.0 + c))).astype(np.float32)

    def make_desc(self, weights=None, grads=None):
        if weights is None:
            weights = [p.weight for p in self.parameter_list]
        theta = self.theta if self.theta.device == weights[0].device else self.theta.to(weights[0].device)
        self.theta = theta.contiguous()
        return K.make_desc(self.kernel_name, list(weights) + [self.theta], None if grads is None else list(grads),
                           tot_entity=self.tot_entity, tot_relation=self.tot_relation, **self.desc_kwargs())


class TransR(PairwiseModel):
    """pairwise.py:367-470.  Normalised entities projected by the relation's [d_e, d_r] matrix, then the TransE tail."""
    kernel_name = "transr"

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(
            ["tot_entity", "tot_relation", "rel_hidden_size", "ent_hidden_size", "l1_flag"], kwargs))
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, self.ent_hidden_size)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, self.rel_hidden_size)
        self.rel_matrix = NamedEmbedding("rel_matrix", self.tot_relation, self.ent_hidden_size * self.rel_hidden_size)
        _xavier(self.ent_embeddings, self.rel_embeddings, self.rel_matrix)
        self.parameter_list = [self.ent_embeddings, self.rel_embeddings, self.rel_matrix]
        self.loss = Criterion.pairwise_hinge

    def desc_kwargs(self):
        return dict(dim=self.ent_hidden_size, rel_dim=self.rel_hidden_size, l1_flag=bool(self.l1_flag))

    def embed(self, h, r, t):
        m = self.rel_matrix(r).view(-1, self.ent_hidden_size, self.rel_hidden_size)
        proj = lambda e: torch.bmm(F.normalize(self.ent_embeddings(e), p=2, dim=-1).unsqueeze(1), m).squeeze(1)
        return proj(h), F.normalize(self.rel_embeddings(r), p=2, dim=-1), proj(t)


class TransH(PairwiseModel):
    """pairwise.py:96-182.  Entities projected onto the relation hyperplane (normal w_r) first."""
    kernel_name = "transh"

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "l1_flag"], kwargs))
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, self.hidden_size)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, self.hidden_size)
        self.w = NamedEmbedding("w", self.tot_relation, self.hidden_size)
        _xavier(self.ent_embeddings, self.rel_embeddings, self.w)
        self.parameter_list = [self.ent_embeddings, self.rel_embeddings, self.w]
        self.loss = Criterion.pairwise_hinge

    def desc_kwargs(self):
        return dict(dim=self.hidden_size, l1_flag=bool(self.l1_flag))

    @staticmethod
    def _projection(emb_e, proj_vec):
        n = F.normalize(proj_vec, p=2, dim=-1)
        return emb_e - (emb_e * n).sum(dim=-1, keepdim=True) * n

    def embed(self, h, r, t):
        w = self.w(r)
        return (self._projection(self.ent_embeddings(h), w), self.rel_embeddings(r),
                self._projection(self.ent_embeddings(t), w))


class TransD(PairwiseModel):
    """pairwise.py:185-278.  e' = e + (e . e_m) r_m  (needs ent_hidden_size == rel_hidden_size)."""
    kernel_name = "transd"

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(
            ["tot_entity", "tot_relation", "rel_hidden_size", "ent_hidden_size", "l1_flag"], kwargs))
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, self.ent_hidden_size)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, self.rel_hidden_size)
        self.ent_mappings = NamedEmbedding("ent_mappings", self.tot_entity, self.ent_hidden_size)
        self.rel_mappings = NamedEmbedding("rel_mappings", self.tot_relation, self.rel_hidden_size)
        _xavier(self.ent_embeddings, self.rel_embeddings, self.ent_mappings, self.rel_mappings)
        self.parameter_list = [self.ent_embeddings, self.rel_embeddings, self.ent_mappings, self.rel_mappings]
        self.loss = Criterion.pairwise_hinge

    def desc_kwargs(self):
        return dict(dim=self.ent_hidden_size, rel_dim=self.rel_hidden_size, l1_flag=bool(self.l1_flag))

    @staticmethod
    def _projection(emb_e, emb_m, proj_vec):
        return emb_e + (emb_e * emb_m).sum(dim=-1, keepdim=True) * proj_vec

    def embed(self, h, r, t):
        r_m = self.rel_mappings(r)
        return (self._projection(self.ent_embeddings(h), self.ent_mappings(h), r_m), self.rel_embeddings(r),
                self._projection(self.ent_embeddings(t), self.ent_mappings(t), r_m))


class RotatE(PairwiseModel):
    """pairwise.py:727-791.  energy = sum |h o r - t|^2 - margin, r = e^{i * rel * pi / range}."""
    kernel_name = "rotate"

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "margin"], kwargs))
        self.embedding_range = (self.margin + 2.0) / self.hidden_size
        self.ent_embeddings = NamedEmbedding("ent_embeddings_real", self.tot_entity, self.hidden_size)
        self.ent_embeddings_imag = NamedEmbedding("ent_embeddings_imag", self.tot_entity, self.hidden_size)
        self.rel_embeddings = NamedEmbedding("rel_embeddings_real", self.tot_relation, self.hidden_size)
        for e in (self.ent_embeddings, self.ent_embeddings_imag, self.rel_embeddings):
            nn.init.uniform_(e.weight, -self.embedding_range, self.embedding_range)
        self.parameter_list = [self.ent_embeddings, self.ent_embeddings_imag, self.rel_embeddings]
        self.loss = Criterion.pariwise_logistic

    def desc_kwargs(self):
        return dict(dim=self.hidden_size, margin=float(self.margin))

    def embed(self, h, r, t):
        phase = self.rel_embeddings(r) / (self.embedding_range / _PI)
        return (self.ent_embeddings(h), self.ent_embeddings_imag(h), torch.cos(phase), torch.sin(phase),
                self.ent_embeddings(t), self.ent_embeddings_imag(t))


class Rescal(PairwiseModel):
    """pairwise.py:794-865.  energy = -h^T M_r t.  Like the reference, every forward/embed first overwrites
    both tables with their row-normalised versions (pairwise.py:843-844)."""
    kernel_name = "rescal"

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(["tot_entity", "tot_relation", "hidden_size", "margin"], kwargs))
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, self.hidden_size)
        self.rel_matrices = NamedEmbedding("rel_matrices", self.tot_relation, self.hidden_size * self.hidden_size)
        _xavier(self.ent_embeddings, self.rel_matrices)
        self.parameter_list = [self.ent_embeddings, self.rel_matrices]
        self.loss = Criterion.pairwise_hinge

    def desc_kwargs(self):
        return dict(dim=self.hidden_size)

    def normalize_tables(self):
        K.rescal_normalize(self.ent_embeddings.weight.data, self.rel_matrices.weight.data, self.hidden_size)

    def forward(self, h, r, t):
        self.normalize_tables()
        return super().forward(h, r, t)

    def embed(self, h, r, t):
        k = self.hidden_size
        self.normalize_tables()
        return (self.ent_embeddings(h).view(-1, k, 1), self.rel_matrices(r).view(-1, k, k),
                self.ent_embeddings(t).view(-1, k, 1))


class NTN(PairwiseModel):
    """pairwise.py:868-963.  energy = -r^ . tanh(h^T W_{1..k} t^ + h^ M1 + t^ M2 + b)."""
    kernel_name = "ntn"

    def __init__(self, **kwargs):
        super().__init__(self.__class__.__name__.lower())
        self.__dict__.update(self.load_params(
            ["tot_entity", "tot_relation", "ent_hidden_size", "rel_hidden_size", "lmbda"], kwargs))
        d, k = self.ent_hidden_size, self.rel_hidden_size
        self.ent_embeddings = NamedEmbedding("ent_embedding", self.tot_entity, d)
        self.rel_embeddings = NamedEmbedding("rel_embedding", self.tot_relation, k)
        self.mr1 = NamedEmbedding("mr1", d, k)
        self.mr2 = NamedEmbedding("mr2", d, k)
        self.br = NamedEmbedding("br", 1, k)
        self.mr = NamedEmbedding("mr", k, d * d)
        _xavier(self.ent_embeddings, self.rel_embeddings, self.mr1, self.mr2, self.br, self.mr)
        self.parameter_list = [self.ent_embeddings, self.rel_embeddings, self.mr1, self.mr2, self.br, self.mr]
        self.loss = Criterion.pairwise_hinge

    def desc_kwargs(self):
        return dict(dim=self.ent_hidden_size, rel_dim=self.rel_hidden_size)

    def embed(self, h, r, t):
        return self.ent_embeddings(h), self.rel_embeddings(r), self.ent_embeddings(t)

    def get_reg(self, h, r, t):
        return self.lmbda * torch.sqrt(sum(torch.sum(p.weight ** 2) for p in self.parameter_list))
