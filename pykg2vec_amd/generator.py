"""Device-resident batch generator (drop-in for pykg2vec/data/generator.py:244-315).

The reference runs one feeder process and `num_process_gen` worker processes that corrupt triples in python loops
and ship lists through multiprocessing queues (data/generator.py:11-158).  Here the train triples, the relation
property (bern) table and an open-addressing hash set of the train triples live in HBM, and a batch is one
`kge_corrupt` kernel launch.  Same iterator surface: `start_one_epoch(num_batch)`, `next()`, `stop()`; the yielded
lists have the reference layouts (pairwise: [ph, pr, pt, nh, nr, nt]; pointwise: [h, r, t, y] with every positive
followed by its neg_rate negatives), as int64 device tensors instead of python lists.

Batch order: like the reference feeder, ONE permutation of the train set is drawn when the generator is created
(data/generator.py:23) and every epoch walks it from the start.
"""
import numpy as np
import torch

from . import kernels as K
from .common import TrainingStrategy


def _triples_array(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.int64)
    if torch.is_tensor(data):
        return data.cpu().numpy().astype(np.int64)
    return np.asarray([[x.h, x.r, x.t] for x in data], dtype=np.int64).reshape(-1, 3)


def relation_property(train, tot_relation):
    """KnowledgeGraph.read_relation_property (data/kgcontroller.py:466-492) on the flat train array:
    prob[r] = |unique tails of r| / (|unique heads of r| + |unique tails of r|), 0 for unseen relations."""
    train = np.asarray(train, dtype=np.int64).reshape(-1, 3)
    E = int(train[:, [0, 2]].max()) + 1 if len(train) else 1
    nh = np.bincount(np.unique(train[:, 1] * E + train[:, 0]) // E, minlength=tot_relation)
    nt = np.bincount(np.unique(train[:, 1] * E + train[:, 2]) // E, minlength=tot_relation)
    tot = nh + nt
    return np.where(tot > 0, nt / np.maximum(tot, 1), 0.0)


def bern_table(prob):
    """float32 device table of the bern probabilities with the reference's comparison preserved exactly: the
    reference tests `np.random.random() > prob` in double precision (data/generator.py:73,77); the device tests a
    24-bit uniform u (a multiple of 2^-24, exact in float32) against table[r].  Rounding each probability TOWARD ZERO
    to float32 keeps every such comparison identical (round-to-nearest could move a probability up past a
    representable u)."""
    p64 = np.asarray(prob, dtype=np.float64)
    p32 = p64.astype(np.float32)
    up = p32.astype(np.float64) > p64
    p32[up] = np.nextafter(p32[up], np.float32(-1.0))
    return p32


class Generator:
    def __init__(self, model, config, seed=None, rank=0, world_size=1, backend=K):
        self.K = backend
        self.model = model
        self.config = config
        self.training_strategy = model.training_strategy
        if self.training_strategy not in (TrainingStrategy.PAIRWISE_BASED, TrainingStrategy.POINTWISE_BASED):
            raise NotImplementedError("This strategy is not supported.")
        self.device = torch.device(config.device if isinstance(config.device, str) else config.device)
        train = _triples_array(config.knowledge_graph.read_cache_data('triplets_train'))
        self.n_train = train.shape[0]
        self.seed = int(seed if seed is not None else getattr(config, "seed", 0) or 0)
        self.rank, self.world_size = rank, world_size
        rng = np.random.default_rng(self.seed)
        perm = rng.permutation(self.n_train)
        # A batch is a SET of triples: inside each batch slice of the permutation the rows are ordered by relation id
        # (once, here), so that the fused kernel can keep a relation row and its gradient in registers across
        # consecutive pairs.  Which triples form which batch is unchanged.
        bsz = int(config.batch_size)
        for lo in range(0, self.n_train, bsz):
            sl = perm[lo:lo + bsz]
            perm[lo:lo + bsz] = sl[np.argsort(train[sl, 1], kind="stable")]
        self.triples = torch.from_numpy(train).to(self.device)
        self.perm = torch.from_numpy(perm).to(self.device)
        self.slots = self.K.triple_set_build(self.triples)
        self.bern = None
        if getattr(config, "sampling", "uniform") == "bern":
            try:
                prop = config.knowledge_graph.read_cache_data('relationproperty')
                table = np.asarray([prop[r] for r in range(config.tot_relation)], dtype=np.float64)
            except (KeyError, FileNotFoundError, AttributeError):  # cache without the pickle: derive it from the split
                table = relation_property(train, config.tot_relation)
            self.bern = torch.from_numpy(bern_table(table)).to(self.device)
        self.neg_rate = int(config.neg_rate)
        self.batch_size = int(config.batch_size)
        self._pending = 0
        self._batch_idx = 0
        self._draws = 0  # Philox counter offset: unique per generated negative over the whole run

    def __iter__(self):
        return self

    def start_one_epoch(self, num_batch):
        self._pending = int(num_batch)
        self._batch_idx = 0

    def stop(self):
        self._pending = 0

    def _next_range(self):
        """(start, n, offset) of the next batch in the permutation / Philox stream; advances the counters."""
        if self._pending <= 0:
            raise StopIteration
        b = self._batch_idx
        self._batch_idx += 1
        self._pending -= 1
        B = self.batch_size
        start, n = B * b, max(0, min(B, self.n_train - B * b))
        offset = self._draws
        self._draws += B * self.neg_rate
        if self.world_size > 1:  # data-parallel shard of the batch; Philox counters stay global
            per = (n + self.world_size - 1) // self.world_size
            lo = min(n, self.rank * per)
            start, n = start + lo, max(0, min(per, n - lo))
            offset += lo * self.neg_rate
        return start, n, offset

    def __next__(self):
        start, n, offset = self._next_range()
        return self.K.sample_batch(self.triples, self.perm, start, n, self.neg_rate, self.config.tot_entity, self.bern,
                                   self.slots, self.seed, offset,
                                   pointwise=self.training_strategy == TrainingStrategy.POINTWISE_BASED)
