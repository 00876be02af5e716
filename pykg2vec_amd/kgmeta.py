"""Model base classes mirroring pykg2vec/models/KGMeta.py:14-80 and models/Domain.py:8-17, with `forward`
routed to the HIP scorer through the custom op `kge::score` (ops.py; dense gradients, like nn.Embedding(sparse=False))."""
import torch
import torch.nn as nn

from . import kernels as K
from . import ops
from .common import TrainingStrategy


class NamedEmbedding(nn.Embedding):
    """nn.Embedding carrying a human-readable `.name` (models/Domain.py:8-17)."""

    def __init__(self, name, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._name = name

    @property
    def name(self):
        return self._name


class Model:
    """KGMeta.Model (models/KGMeta.py:14-38)."""

    kernel_name = None  # key into kernels.MODEL_IDS

    def load_params(self, param_list, kwargs):
        for param_name in param_list:
            if param_name not in kwargs:
                raise Exception("hyperparameter %s not found!" % param_name)
            self.database[param_name] = kwargs[param_name]
        return self.database

    def get_reg(self, h, r, t, **kwargs):
        return 0.0

    # ---- HIP plumbing
    def desc_kwargs(self):
        raise NotImplementedError

    def make_desc(self, weights=None, grads=None):
        if weights is None:
            weights = [p.weight for p in self.parameter_list]
        return K.make_desc(self.kernel_name, list(weights), None if grads is None else list(grads),
                           tot_entity=self.tot_entity, tot_relation=self.tot_relation, **self.desc_kwargs())

    def forward(self, h, r, t):
        """Energies of the triples (h, r, t) through the dispatcher-registered op `kge::score` (ops.py): differentiable w.r.t. the
        tables (dense gradients, like nn.Embedding(sparse=False)), visible to torch.ops / torch.compile."""
        return torch.ops.kge.score(self._kge_op_key, h, r, t, [p.weight for p in self.parameter_list])

    def _register_op_handle(self):
        """The integer handle torch.ops.kge.score finds this model by: taken once per object (construction, unpickling, deepcopy),
        so that forward() reads a plain attribute -- a constant for torch.compile -- instead of touching the weak registry."""
        self.__dict__.pop("_kge_op_key", None)
        ops.register_model(self)

    def _restore(self, state):   # unpickling / deepcopy: the copy is a new object and needs its own handle
        nn.Module.__setstate__(self, state)
        self._register_op_handle()

    # ---- the reference Evaluator's optional hooks (utils/evaluator.py:250-252,263-265): candidate ids by
    # descending energy, shape [1, topk]; served by the sweep kernels instead of forward() over E id tensors.
    def _sweep(self, h, r, t, side):
        """float32 [E]: energies of (h, r, e) (side 0) or of (e, r, t) (side 1) for every entity e -- ONE sweep (round 5 ran both
        and threw one away); the id of the swept position is a placeholder the kernel never reads."""
        h0, r0, t0 = h.view(-1)[0], r.view(-1)[0], t.view(-1)[0]
        trip = torch.stack([h0, r0, t0]).view(1, 3).contiguous()
        return K.eval_sweep_scores_side(self.make_desc(), trip, side)[0]

    def predict_tail_rank(self, h, r, topk=-1):
        _, rank = torch.topk(self._sweep(h, r, torch.zeros_like(h), 0), k=topk)
        return rank.view(1, -1)

    def predict_head_rank(self, t, r, topk=-1):
        _, rank = torch.topk(self._sweep(torch.zeros_like(t), r, t, 1), k=topk)
        return rank.view(1, -1)


class PairwiseModel(nn.Module, Model):
    forward = Model.forward  # nn.Module.forward precedes Model.forward in the MRO
    __setstate__ = Model._restore

    def __init__(self, model_name):
        super().__init__()
        self.model_name = model_name
        self.training_strategy = TrainingStrategy.PAIRWISE_BASED
        self.database = {}
        self._register_op_handle()


class PointwiseModel(nn.Module, Model):
    forward = Model.forward
    __setstate__ = Model._restore

    def __init__(self, model_name):
        super().__init__()
        self.model_name = model_name
        self.training_strategy = TrainingStrategy.POINTWISE_BASED
        self.database = {}
        self._register_op_handle()
