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


def _next_pow2(x):
    x = np.asarray(x, dtype=np.int64)
    return np.where(x <= 1, 1, 1 << np.ceil(np.log2(np.maximum(x, 1))).astype(np.int64))


def build_pull_batch(pos, tot_entity, tot_relation, segment, groups_per_block=8, compact=False):
    """Incidence index of ONE batch for the owner-computes step (csrc/kge_pull.hip): every parameter row (entities first,
    then tot_entity + relation) gets the sorted list of the (pair, role) slots it occupies in the batch -- role 0 = head,
    1 = tail, 2 = relation -- cut into work items of at most `segment` incidences.

    Item kinds: 0 = the row's only item; 3 = one of 2..groups_per_block items of a row, all placed in consecutive owner
    groups of ONE workgroup (they combine their partial sums through LDS, in segment order); 1 / 2 = first / later item
    of a row with more items than that (partial sums through global memory + the finishing kernel).  Items are laid out
    in workgroup slots (groups_per_block per workgroup = kge_pull_groups_per_block(dim); padding items have row -1).

    Placement -- THE rule, restated by the device builder (csrc/kge_index.hip: kge_pull_index_build produces the same arrays):
      region A  rows of 2..groups_per_block items, ordered by (p2 descending, row ascending) with p2 = the item count rounded up to
                a power of two; a row's items occupy the first slots of an aligned run of p2 slots (descending powers of two
                keep every run inside one workgroup); the region is rounded up to whole workgroups;
      region B  every other item (single-item rows, items of rows with more than groups_per_block items), ordered by
                (weight descending, row ascending, segment ascending) with weight = incidences (+ ~B/E for an entity row's
                first item, which also walks the row's corrupting-entity draws): the q-th of them takes the q-th OPEN slot --
                first the slots region A left free, in slot order, then the slots after region A.
    Heavy items therefore start first and the lightest workgroups run last.
    compact=True lists only the rows that have an incidence and returns, as a sixth value, the bitmap of those rows (int32
    words): the kernel visits every other row implicitly after the listed items (small batches of a big graph touch a small
    part of the tables, and an explicit item per untouched row and batch would dominate the index).
    Returns int32 arrays (pairs [B,4], inc [3B], items [n_slots,4], multi [n_multi,4]) and the number of partial slots."""
    pos = np.asarray(pos, dtype=np.int64).reshape(-1, 3)
    B, E, nrows = len(pos), int(tot_entity), int(tot_entity) + int(tot_relation)
    GPB, seg = int(groups_per_block), int(segment)
    i = np.arange(B, dtype=np.int64)
    rows = np.concatenate([pos[:, 0], pos[:, 2], E + pos[:, 1]])
    vals = np.concatenate([4 * i, 4 * i + 1, 4 * i + 2])
    order = np.lexsort((vals, rows))
    inc = vals[order].astype(np.int32)
    counts = np.bincount(rows, minlength=nrows)
    row_off = np.cumsum(counts) - counts
    # ---- the row list: every row, or (compact) the rows with an incidence, ascending
    listed = np.flatnonzero(counts > 0) if compact else np.arange(nrows, dtype=np.int64)
    beg, cnt = row_off[listed], counts[listed]
    nseg = np.maximum(1, (cnt + seg - 1) // seg)
    is_local = (nseg >= 2) & (nseg <= GPB)
    is_global = nseg > GPB
    c_extra = max(1, B // max(E, 1))                        # entity owners also walk ~B/E corrupting-entity draws
    # ---- region A
    lu = np.flatnonzero(is_local)
    p2 = _next_pow2(nseg[lu])
    lu_order = np.lexsort((listed[lu], -p2))
    lu, p2 = lu[lu_order], p2[lu_order]
    a_start = np.cumsum(p2) - p2
    SA = int(p2.sum())
    SA_pad = (SA + GPB - 1) // GPB * GPB
    # free slots of region A, in slot order
    free_n = p2 - nseg[lu]
    open_slots = np.concatenate([np.repeat(a_start + nseg[lu], free_n) + (np.arange(int(free_n.sum())) - np.repeat(np.cumsum(free_n) - free_n, free_n)),
                                 np.arange(SA, SA_pad, dtype=np.int64)]).astype(np.int64)
    F = len(open_slots)
    # ---- region B items: (list index u, segment s)
    nb_items = np.where(is_local, 0, nseg)
    bu = np.repeat(np.arange(len(listed), dtype=np.int64), nb_items)
    bs = np.arange(int(nb_items.sum()), dtype=np.int64) - np.repeat(np.cumsum(nb_items) - nb_items, nb_items)
    b_beg = beg[bu] + bs * seg
    b_end = np.minimum(b_beg + seg, beg[bu] + cnt[bu])
    weight = (b_end - b_beg) + np.where((listed[bu] < E) & (bs == 0), c_extra, 0)
    g_base = np.cumsum(np.where(is_global, nseg, 0)) - np.where(is_global, nseg, 0)     # partial slots, consecutive per row in row order
    b_info = np.where(is_global[bu], np.where(bs == 0, 1, 2) | ((g_base[bu] + bs) << 2), 0)
    b_order = np.lexsort((bs, listed[bu], -weight))
    NB = len(bu)
    q = np.arange(NB, dtype=np.int64)
    b_slot = np.where(q < F, open_slots[np.minimum(q, max(F - 1, 0))] if F else 0, SA_pad + (q - F))
    n_slots = max(SA_pad, SA_pad + max(0, NB - F))
    n_slots = max(GPB, (n_slots + GPB - 1) // GPB * GPB)
    out = np.zeros((n_slots, 4), dtype=np.int64)
    out[:, 0] = -1
    if NB:
        o = b_order
        out[b_slot] = np.stack([listed[bu][o], b_beg[o], b_end[o], b_info[o]], 1)
    if len(lu):
        au = np.repeat(np.arange(len(lu), dtype=np.int64), nseg[lu])
        a_s = np.arange(int(nseg[lu].sum()), dtype=np.int64) - np.repeat(np.cumsum(nseg[lu]) - nseg[lu], nseg[lu])
        u = lu[au]
        a_beg = beg[u] + a_s * seg
        out[a_start[au] + a_s] = np.stack([listed[u], a_beg, np.minimum(a_beg + seg, beg[u] + cnt[u]), 3 | (a_s << 2) | (nseg[u] << 6)], 1)
    gu = np.flatnonzero(is_global)
    multi = np.stack([listed[gu], g_base[gu], nseg[gu], np.zeros_like(gu)], 1).astype(np.int32).reshape(-1, 4)
    pairs = np.concatenate([pos, np.zeros((B, 1), np.int64)], 1).astype(np.int32)
    res = (pairs, inc, np.ascontiguousarray(out.astype(np.int32)), np.ascontiguousarray(multi), int(np.where(is_global, nseg, 0).sum()))
    if compact:
        bits = np.zeros((nrows + 31) // 32 * 32, dtype=np.uint8)
        bits[:nrows] = counts > 0
        words = np.packbits(bits.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
        res += (words.view(np.int32),)
    return res


class PullIndex:
    """Device-resident incidence index of every batch of the epoch (a batch is a fixed slice of the generator's
    permutation, data/generator.py:23-35, so the index is built once).  Two builders with identical output:
    `PullIndex(batches, ...)` -- numpy, for explicit one-off batches and CPU tests -- and `PullIndex.build_on_device(...)`
    -- csrc/kge_index.hip, all batches of the epoch order in a handful of launches (the product path)."""

    SEGMENT = 8  # incidences per work item: rows with longer lists are cut up and finished by a second small kernel

    @staticmethod
    def _segment(segment, groups_per_block):
        import os
        seg = int(segment or os.environ.get("KGE_PULL_SEGMENT") or PullIndex.SEGMENT)   # env: tuning sweeps only
        group = 256 // int(groups_per_block)   # lanes of an owner group: one visit descriptor per lane
        if not 1 <= seg <= group:
            raise ValueError("PullIndex: segment %d must be in [1, %d] (incidences of a work item map one per lane)" % (seg, group))
        return seg

    @staticmethod
    def _compact_rule(n_pairs, nrows):
        """A batch touches at most 3 B rows: list only those when that is the smaller index."""
        return 3 * n_pairs * 2 <= nrows

    def __init__(self, batches, tot_entity, tot_relation, device, segment=None, groups_per_block=8, compact=None):
        seg = self._segment(segment, groups_per_block)
        nrows = int(tot_entity) + int(tot_relation)
        if compact is None:
            compact = bool(batches) and self._compact_rule(len(batches[0]), nrows)
        self.compact = bool(compact)
        self.segment, self.built_on = seg, "host"
        built = [build_pull_batch(b, tot_entity, tot_relation, seg, groups_per_block, compact=self.compact) for b in batches]
        self.words = (nrows + 31) // 32
        self.n_batches = len(built)
        self.batch_size = len(batches[0]) if built else 0
        self.max_slots = max([x[4] for x in built] + [1])
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
        self._views = [tuple(dev(x[k]) for k in range(4)) for x in built]
        self._skips = [dev(x[5]) for x in built] if self.compact else None
        self._invs = []
        for x in built:   # inverse of the sorted incidence list: inv[3 * pair + role] = its position (what the sampler files by)
            inc = x[1].astype(np.int64)
            inv = np.empty(len(inc), dtype=np.int32)
            inv[3 * (inc >> 2) + (inc & 3)] = np.arange(len(inc), dtype=np.int32)
            self._invs.append(dev(inv))

    @classmethod
    def build_on_device(cls, backend, triples, perm, n_batches, batch_stride, slice_lo, n_pairs, tot_entity, tot_relation,
                        segment=None, groups_per_block=8, compact=None):
        """All `n_batches` batches at once on the device: batch b = triples[perm[b * batch_stride + slice_lo + i]], i < n_pairs."""
        self = cls.__new__(cls)
        seg = cls._segment(segment, groups_per_block)
        nrows = int(tot_entity) + int(tot_relation)
        self.compact = bool(cls._compact_rule(n_pairs, nrows) if compact is None else compact)
        self.segment, self.built_on = seg, "device"
        self.words = (nrows + 31) // 32
        self.n_batches, self.batch_size = int(n_batches), int(n_pairs)
        out = backend.pull_index_build(triples, perm, batch_stride, slice_lo, n_pairs, n_batches, tot_entity, tot_relation, seg,
                                       groups_per_block, self.compact)
        pairs, inc, inv, items, multi, skip, counts = out
        self._invs = [inv[b] for b in range(self.n_batches)]
        self._storage = out
        cnt = counts.cpu().numpy().reshape(-1, 4)      # the one host read of the build: live slots / rows per batch
        self.max_slots = int(max(1, cnt[:, 2].max())) if len(cnt) else 1
        self._views = [(pairs[b], inc[b], items[b, :int(cnt[b, 0])], multi[b, :int(cnt[b, 1])]) for b in range(self.n_batches)]
        self._skips = [skip[b] for b in range(self.n_batches)] if self.compact else None
        return self

    def inv(self, b):
        """Inverse of batch b's sorted incidence list: inv[3 * pair + role] = position in `inc`."""
        return self._invs[b]

    def skip(self, b):
        """Bitmap (int32 words) of the rows batch b lists explicitly, or None when its items cover every row."""
        return self._skips[b] if self._skips is not None else None

    @staticmethod
    def footprint(backend, n_batches, batch_size, tot_entity, tot_relation, segment=None, groups_per_block=8, compact=None):
        """(resident bytes, peak bytes while building) of the device-built index of `n_batches` batches plus the two per-step
        sampler list sets every owner-computes path keeps: the exact strides of kge_pull_index_geometry (the device builder sizes
        every batch's arrays for the worst case), not an estimate."""
        seg = PullIndex._segment(segment, groups_per_block)
        nrows = int(tot_entity) + int(tot_relation)
        if compact is None:
            compact = PullIndex._compact_rule(batch_size, nrows)
        resident, workspace = backend.pull_index_bytes(n_batches, batch_size, tot_entity, tot_relation, seg, groups_per_block, compact)
        resident += 2 * backend.pull_list_set_bytes(batch_size, tot_entity)
        return resident, resident + workspace

    def batch(self, b):
        """(pairs, inc, items, multi) views of batch b; incidence / pair indices inside are relative to the batch."""
        return self._views[b]


class StagedIndex:
    """Static incidence CSR of every batch for the staged (atomic-free) bundle step (kge_optimizer_step_staged): per batch,
    for every entity the positives it heads / tails (entries positive << 1 | side, ascending) and for every relation the
    positives it labels.  Batches are fixed slices of the generator's permutation, so this is built once."""

    MAX_BYTES = 1 << 30
    REL_CHUNK = 16      # csrc/kge_staged.hip kRelChunk
    LONG_LIST = 64      # relation lists beyond this are pre-reduced in chunks by many waves

    def __init__(self, batches, tot_entity, tot_relation, device):
        E, R = int(tot_entity), int(tot_relation)
        ent_off, ent_inc, rel_off, rel_inc, chunk_off, chunk_rel, t_ent, t_rel = [], [], [], [], [], [], [], []
        self.max_rel_list = 0
        for b in batches:
            n = len(b)
            i = np.arange(n, dtype=np.int64)
            ent = np.concatenate([b[:, 0], b[:, 2]])
            x = np.concatenate([i * 2, i * 2 + 1])
            order = np.lexsort((x, ent))
            ent_inc.append(x[order].astype(np.int32))
            ent_off.append(np.concatenate([[0], np.cumsum(np.bincount(ent, minlength=E))]).astype(np.int32))
            rel_inc.append(np.argsort(b[:, 1], kind="stable").astype(np.int32))
            cnt = np.bincount(b[:, 1], minlength=R)
            rel_off.append(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32))
            nch = (cnt + self.REL_CHUNK - 1) // self.REL_CHUNK
            chunk_off.append(np.concatenate([[0], np.cumsum(nch)]).astype(np.int32))
            chunk_rel.append(np.repeat(np.arange(R, dtype=np.int32), nch))
            self.max_rel_list = max(self.max_rel_list, int(cnt.max()) if n else 0)
            t_ent.append(np.unique(ent).astype(np.int32))
            t_rel.append(np.flatnonzero(cnt).astype(np.int32))
        self.sizes = [len(b) for b in batches]
        self.n_chunks = [len(x) for x in chunk_rel]
        self.chunk_pos = np.concatenate([[0], np.cumsum(self.n_chunks)]).astype(np.int64)
        self.pos_off = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.E, self.R = E, R

        def cat(xs):
            return torch.from_numpy(np.concatenate(xs)).to(device)

        self.ent_off, self.ent_inc, self.rel_off, self.rel_inc = cat(ent_off), cat(ent_inc), cat(rel_off), cat(rel_inc)
        self.chunk_off, self.chunk_rel = cat(chunk_off), cat(chunk_rel + [np.zeros(1, np.int32)])
        self.t_ent, self.t_rel = cat(t_ent + [np.zeros(1, np.int32)]), cat(t_rel + [np.zeros(1, np.int32)])
        self.t_ent_pos = np.concatenate([[0], np.cumsum([len(x) for x in t_ent])]).astype(np.int64)
        self.t_rel_pos = np.concatenate([[0], np.cumsum([len(x) for x in t_rel])]).astype(np.int64)

    @classmethod
    def fits(cls, n_batches, tot_entity, tot_relation):
        return n_batches * (tot_entity + tot_relation + 2) * 4 <= cls.MAX_BYTES

    def batch(self, b):
        E, R, p = self.E, self.R, self.pos_off
        return (self.ent_off[b * (E + 1):(b + 1) * (E + 1)], self.ent_inc[2 * p[b]:2 * p[b + 1]],
                self.rel_off[b * (R + 1):(b + 1) * (R + 1)], self.rel_inc[p[b]:p[b + 1]], self.sizes[b])

    def touched(self, b):
        """(entities, n, relations, n) taking part in batch b's positives: the static part of the sparse sweep's work list."""
        a, r = self.t_ent_pos, self.t_rel_pos
        return (self.t_ent[a[b]:max(a[b + 1], a[b] + 1)], int(a[b + 1] - a[b]),
                self.t_rel[r[b]:max(r[b + 1], r[b] + 1)], int(r[b + 1] - r[b]))

    def chunks(self, b):
        """(rel_chunk_off, chunk_rel, n_chunks) of batch b, or None when no relation list is long enough to pay."""
        if self.max_rel_list <= self.LONG_LIST:
            return None
        R, c = self.R, self.chunk_pos
        return self.chunk_off[b * (R + 1):(b + 1) * (R + 1)], self.chunk_rel[c[b]:max(c[b + 1], c[b] + 1)], self.n_chunks[b]


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
        self._train_np, self._perm_np = train, perm
        self._pull_index = None
        self._pending = 0
        self._batch_idx = 0
        self._draws = 0  # Philox counter offset: unique per generated negative over the whole run

    def pull_index(self, groups_per_block=None, compact=None, segment=None):
        """Incidence index of every FULL batch of the permutation (built on first use).  groups_per_block: owner groups per
        workgroup of the consuming kernel (default: kge_pull_step's); compact: force / forbid the touched-rows-only form;
        segment: incidences per work item (default: `self.pull_segment` if the trainer set one, else PullIndex.SEGMENT)."""
        segment = segment or getattr(self, "pull_segment", None)
        key = (groups_per_block, compact, segment)
        if self._pull_index is not None and getattr(self, "_pull_index_key", key) != key:
            self._pull_index = None
        if self._pull_index is None:
            self._pull_index_key = key
            B = self.batch_size
            nb = self.n_train // B
            gpb = groups_per_block or self.K.pull_groups_per_block(self.model.hidden_size)
            per, lo = B, 0
            if self.world_size > 1:   # data-parallel rank: its slice of every batch (the rule of _next_range; B % world == 0)
                per = (B + self.world_size - 1) // self.world_size
                lo = self.rank * per
            import time
            t0 = time.perf_counter()
            if hasattr(self.K, "pull_index_build") and self.triples.is_cuda and nb > 0:
                # the product path: every batch of the epoch order indexed on the device (csrc/kge_index.hip)
                self._pull_index = PullIndex.build_on_device(self.K, self.triples, self.perm, nb, B, lo, per, self.config.tot_entity,
                                                             self.config.tot_relation, segment=segment, groups_per_block=gpb,
                                                             compact=compact)
            else:   # no device (CPU tests with an injected backend): the numpy restatement of the same rule
                pos = self._train_np[self._perm_np[:nb * B]].reshape(nb, B, 3)[:, lo:lo + per]
                self._pull_index = PullIndex(list(pos), self.config.tot_entity, self.config.tot_relation, self.device, segment,
                                             groups_per_block=gpb, compact=compact)
            self.pull_index_ms = (time.perf_counter() - t0) * 1e3   # set-up cost of the owner-computes path (host wall, incl. the sync)
        return self._pull_index

    def staged_index(self):
        """Static incidence CSR of every batch of the permutation, the short last one included (built on first use)."""
        if getattr(self, "_staged_index", None) is None:
            B = self.batch_size
            pos = self._train_np[self._perm_np]
            batches = [pos[lo:lo + B] for lo in range(0, self.n_train, B)]
            self._staged_index = StagedIndex(batches, self.config.tot_entity, self.config.tot_relation, self.device)
        return self._staged_index

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
