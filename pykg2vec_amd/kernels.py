"""Torch-tensor front end of the C ABI: argument checks, descriptor building, stream plumbing.

PyTorch is used here only for device memory and the current HIP stream; all arithmetic happens in
libkge_hip.so.  Every function raises if a tensor is not a contiguous tensor on a HIP device.
"""
import ctypes
import os
import math

import torch

from . import _lib as L

MODEL_IDS = {"transe": L.TRANSE, "transh": L.TRANSH, "transd": L.TRANSD, "rotate": L.ROTATE, "rescal": L.RESCAL,
             "ntn": L.NTN, "distmult": L.DISTMULT, "complex": L.COMPLEX, "complexn3": L.COMPLEX, "analogy": L.ANALOGY,
             "transm": L.TRANSM, "cp": L.CP, "simple": L.SIMPLE, "simple_ignr": L.SIMPLE_IGNR, "quate": L.QUATE,
             "transr": L.TRANSR}
OPTIMIZER_IDS = {"sgd": L.OPT_SGD, "adam": L.OPT_ADAM, "adagrad": L.OPT_ADAGRAD, "rms": L.OPT_RMSPROP,
                 "gradient": 4}   # KGE_OPT_GRADIENT: kge_pull_step writes the dense gradient instead of updating


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, what):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a tensor" % what)
    if not t.is_cuda:
        raise L.KgeHipError("%s must live on the HIP device (got %s); the HIP path has no CPU fallback" % (what, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % what)
    return ctypes.c_void_p(t.data_ptr())


def _ids(t, what):
    return _dev(t, torch.int64, what)


# Shape every table of a model must have for the ids the kernels index it with: (rows indexed by, columns).  "E" / "R": the
# kernels read row `entity id` / `relation id`, so the table needs at least tot_entity / tot_relation rows; an int: exactly that
# many rows.  d = dim, k = rel_dim.  (models/pairwise.py, models/pointwise.py constructors; include/kge_hip.h enum kge_model.)
_TABLE_SHAPES = {
    "transe": [("E", "d"), ("R", "d")],
    "transm": [("E", "d"), ("R", "d"), ("R", None)],                      # theta [R]
    "transh": [("E", "d"), ("R", "d"), ("R", "d")],
    "transd": [("E", "d"), ("R", "k"), ("E", "d"), ("R", "k")],
    "rotate": [("E", "d"), ("E", "d"), ("R", "d")],
    "rescal": [("E", "d"), ("R", "d*d")],
    "ntn": [("E", "d"), ("R", "k"), ("d", "k"), ("d", "k"), (1, "k"), ("k", "d*d")],
    "transr": [("E", "d"), ("R", "k"), ("R", "d*k")],
    "distmult": [("E", "d"), ("R", "d")],
    "complex": [("E", "d"), ("E", "d"), ("R", "d"), ("R", "d")],
    "complexn3": [("E", "d"), ("E", "d"), ("R", "d"), ("R", "d")],
    "analogy": [("E", "d"), ("R", "d"), ("E", "d/2"), ("E", "d/2"), ("R", "d/2"), ("R", "d/2")],
    "cp": [("E", "d"), ("R", "d"), ("E", "d")],
    "simple": [("E", "d"), ("E", "d"), ("R", "d"), ("R", "d")],
    "simple_ignr": [("E", "d"), ("E", "d"), ("R", "d"), ("R", "d")],
    # QuatE's four relation tables are allocated with tot_entity rows (models/pointwise.py:622-631) but looked up by RELATION
    # id: tot_relation > tot_entity is an IndexError in the reference and a refused descriptor here
    "quate": [("E", "d")] * 4 + [("R", "d")] * 4,
}


def _check_table_shapes(model_name, tables, tot_entity, tot_relation, dim, rel_dim):
    spec = _TABLE_SHAPES.get(model_name)
    if spec is None:
        return
    if len(tables) < len(spec):
        raise L.KgeHipError("%s: %d tables given, the model has %d" % (model_name, len(tables), len(spec)))
    size = {"E": int(tot_entity), "R": int(tot_relation), "d": int(dim), "k": int(rel_dim)}
    cols_of = {"d": size["d"], "k": size["k"], "d*d": size["d"] ** 2, "d*k": size["d"] * size["k"], "d/2": size["d"] // 2}
    for i, ((by, cols), t) in enumerate(zip(spec, tables)):
        rows = t.shape[0]
        want_cols = cols_of.get(cols)
        got_cols = t.numel() // rows if rows else 0
        if by in ("E", "R"):
            need = size[by]
            if rows < need:
                raise L.KgeHipError(
                    "%s: table %d has %d rows but is indexed by %s ids up to %d (tot_%s = %d): an nn.Embedding lookup would raise "
                    "IndexError (models/Domain.py:8-13)" % (model_name, i, rows, "entity" if by == "E" else "relation", need - 1,
                                                             "entity" if by == "E" else "relation", need))
        else:
            need = by if isinstance(by, int) else size[by]
            if rows != need:
                raise L.KgeHipError("%s: table %d must have %d rows (got %d)" % (model_name, i, need, rows))
        if want_cols is not None and (t.dim() != 2 or got_cols != want_cols):
            raise L.KgeHipError("%s: table %d must be [rows, %d] (got %s)" % (model_name, i, want_cols, tuple(t.shape)))


def set_debug(check_ids=True):
    """Debug mode of the library (include/kge_hip.h: kge_set_debug; KGE_DEBUG_IDS=1 in the environment does the same): every id
    handed to an entry point is range-checked against its table first and a bad one raises KgeHipError, as nn.Embedding raises
    IndexError for the reference.  Synchronises the stream on every call: not for the hot loop."""
    L.load().kge_set_debug(1 if check_ids else 0)


def set_switch(name, value):
    """Force (0 / 1 / an integer) or release (None) one of the library's A/B switches (kge_set_switch; e.g. "EVAL_GEMM")."""
    L.check(L.load().kge_set_switch(name.encode(), -1 if value is None else int(value)), "kge_set_switch")


def debug_enabled():
    return bool(L.load().kge_get_debug())


def check_ids(ids, bound, what="ids"):
    """Raise KgeHipError if some id is outside [0, bound) (one scan kernel + a stream synchronisation)."""
    L.check(L.load().kge_check_ids(_ids(ids, what), ids.numel(), int(bound), _stream()), "kge_check_ids(%s)" % what)


def debug_marker(tag):
    """An empty launch of `tag` workgroups ("kge::k_marker"): a cut point in a profiler's dispatch sequence (bench.py)."""
    L.check(L.load().kge_debug_marker(int(tag), _stream()), "kge_debug_marker")


def make_desc(model_name, tables, grads=None, *, tot_entity, tot_relation, dim, rel_dim=None, l1_flag=False,
              margin=0.0):
    """Build a kge_model_desc.  `tables` / `grads`: lists of fp32 device tensors in parameter_list order.  Table shapes are
    checked against (tot_entity, tot_relation, dim, rel_dim): the kernels index rows by id without a bounds test."""
    _check_table_shapes(model_name, tables, tot_entity, tot_relation, dim, rel_dim if rel_dim is not None else dim)
    d = L.ModelDesc()
    d.model = MODEL_IDS[model_name]
    d.flags = L.FLAG_L1 if l1_flag else 0
    d.tot_entity, d.tot_relation = int(tot_entity), int(tot_relation)
    d.dim = int(dim)
    d.rel_dim = int(rel_dim if rel_dim is not None else dim)
    d.margin = float(margin)
    d.phase_scale = 0.0
    if model_name == "rotate":
        # embedding_range = (margin + 2) / hidden_size ; phase = r / (embedding_range / pi)   (pairwise.py:747,781)
        d.phase_scale = 1.0 / (((float(margin) + 2.0) / dim) / 3.14159265358979323846)
    for i, t in enumerate(tables):
        d.tables[i] = _dev(t, torch.float32, "table %d" % i).value
    if grads is not None:
        for i, g in enumerate(grads):
            if g.shape != tables[i].shape:
                raise ValueError("grad %d shape %s != table shape %s" % (i, tuple(g.shape), tuple(tables[i].shape)))
            d.grads[i] = _dev(g, torch.float32, "grad %d" % i).value
    d._keepalive = (tables, grads)  # python-side references
    return d


def model_desc(model, weights=None, grads=None):
    """Descriptor of a drop-in model object (kgmeta.Model.make_desc)."""
    return model.make_desc(weights, grads)


def _workspace(desc, n, device):
    """(ptr, nbytes, keepalive) scratch for a score/train call on n rows; (None, 0, None) for gather-type models."""
    nbytes = L.load().kge_workspace_bytes(ctypes.byref(desc), int(n))
    if nbytes == 0:
        return None, 0, None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ctypes.c_void_p(ws.data_ptr()), nbytes, ws


def score_forward(desc, h, r, t):
    n = h.numel()
    if r.numel() != n or t.numel() != n:
        raise ValueError("h, r, t must have equal lengths")
    out = torch.empty(n, dtype=torch.float32, device=h.device)
    wp, wb, _keep = _workspace(desc, n, h.device)
    L.check(L.load().kge_score_forward(ctypes.byref(desc), _ids(h, "h"), _ids(r, "r"), _ids(t, "t"), n,
                                       _dev(out, torch.float32, "scores"), wp, wb, _stream()), "kge_score_forward")
    return out


def score_backward(desc, h, r, t, dscore):
    n = h.numel()
    wp, wb, _keep = _workspace(desc, n, h.device)
    L.check(L.load().kge_score_backward(ctypes.byref(desc), _ids(h, "h"), _ids(r, "r"), _ids(t, "t"), n,
                                        _dev(dscore, torch.float32, "dscore"), wp, wb, _stream()), "kge_score_backward")


_rescal_scratch = {}


def rescal_normalize(ent, rel, k):
    lib = L.load()
    need = lib.kge_rescal_normalize_scratch_bytes(rel.shape[0], int(k))
    key = (ent.device, need)
    if key not in _rescal_scratch:   # a few hundred floats, kept per (device, size): the call stays allocation-free and capturable
        _rescal_scratch[key] = torch.empty(max(1, need // 4), dtype=torch.float32, device=ent.device)
    sc = _rescal_scratch[key]
    L.check(lib.kge_rescal_normalize_ws(_dev(ent, torch.float32, "ent"), ent.shape[0], _dev(rel, torch.float32, "rel"),
                                        rel.shape[0], k, sc.data_ptr(), sc.numel() * 4, _stream()), "kge_rescal_normalize_ws")


def new_loss_buffer(device):
    return torch.zeros(L.LOSS_SLOTS * L.LOSS_STRIDE, dtype=torch.float32, device=device)


def read_loss(buf):
    """Total of the striped loss accumulators (a device tensor; no sync)."""
    return buf.view(L.LOSS_SLOTS, L.LOSS_STRIDE)[:, 0].sum()


def train_pairwise_hinge(desc, ph, pr, pt, nh, nr, nt, margin, loss_buf):
    n = ph.numel()
    if nh.numel() != n:
        raise ValueError("pairwise_hinge needs neg_rate == 1 (criterion.py:27 adds [B] to [B*neg_rate])")
    wp, wb, _keep = _workspace(desc, n, ph.device)
    L.check(L.load().kge_train_pairwise_hinge(ctypes.byref(desc), _ids(ph, "ph"), _ids(pr, "pr"), _ids(pt, "pt"),
                                              _ids(nh, "nh"), _ids(nr, "nr"), _ids(nt, "nt"), n, float(margin),
                                              wp, wb, _dev(loss_buf, torch.float32, "loss"), _stream()),
            "kge_train_pairwise_hinge")


def train_pairwise_hinge_sampled(desc, triples, perm, start, n, bern_prob, slots, seed, offset, margin, loss_buf,
                                 cursor=None):
    """Sampler + both scores + hinge + backward in ONE launch; the batch equals sample_batch(start, n, neg_rate=1)."""
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    pc = _dev(cursor, torch.int64, "cursor") if cursor is not None else None
    L.check(L.load().kge_train_pairwise_hinge_sampled(ctypes.byref(desc), _ids(triples, "triples"), _ids(perm, "perm"),
                                                      int(start), int(n), bp, sp,
                                                      slots.numel() if slots is not None else 0,
                                                      int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), pc,
                                                      float(margin), _dev(loss_buf, torch.float32, "loss"), _stream()),
            "kge_train_pairwise_hinge_sampled")


def train_pointwise_logistic_sampled(desc, triples, perm, start, n_pos, neg_rate, bern_prob, slots, seed, offset, lmbda,
                                     reg_type, loss_buf, cursor=None):
    """Sampler + scoring + pointwise logistic loss + regulariser + backward in ONE launch; the rows equal
    sample_batch(start, n_pos, neg_rate, pointwise=True)."""
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    pc = _dev(cursor, torch.int64, "cursor") if cursor is not None else None
    L.check(L.load().kge_train_pointwise_logistic_sampled(ctypes.byref(desc), _ids(triples, "triples"), _ids(perm, "perm"),
                                                          int(start), int(n_pos), int(neg_rate), bp, sp,
                                                          slots.numel() if slots is not None else 0,
                                                          int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), pc,
                                                          float(lmbda), int(reg_type), _dev(loss_buf, torch.float32, "loss"),
                                                          _stream()), "kge_train_pointwise_logistic_sampled")


def train_pairwise_selfadv_sampled(desc, triples, perm, start, n_pos, neg_rate, alpha, bern_prob, slots, seed, offset,
                                   loss_buf, cursor=None):
    """RotatE: sampler + scores + self-adversarial loss + backward in ONE launch (batch == sample_batch(...))."""
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    pc = _dev(cursor, torch.int64, "cursor") if cursor is not None else None
    L.check(L.load().kge_train_pairwise_selfadv_sampled(ctypes.byref(desc), _ids(triples, "triples"), _ids(perm, "perm"),
                                                        int(start), int(n_pos), int(neg_rate), float(alpha), bp, sp,
                                                        slots.numel() if slots is not None else 0,
                                                        int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), pc,
                                                        _dev(loss_buf, torch.float32, "loss"), _stream()),
            "kge_train_pairwise_selfadv_sampled")


STAGED_CAP = 16   # bucket entries per corrupting entity before the overflow chain


class StagedPlan:
    """struct kge_staged_step for one model: flat parameter / state buffers, the table list with its slot sites, the
    staging buffer and the per-entity registration lists.  `bind(batch_index_views, n_pos)` points it at one batch."""

    LAYOUTS = {   # model -> (static slots per positive, dynamic slots per negative, [(cls, site_a, site_b, dsite)] per table)
        "rotate": (5, 2, [(0, 0, 3, 0), (0, 1, 4, 1), (1, 2, 0, -1)]),                      # h_re h_im r t_re t_im | c_re c_im
        "distmult": (3, 1, [(0, 0, 2, 0), (1, 1, 0, -1)]),                                   # h r t | c
        "complex": (6, 2, [(0, 0, 4, 0), (0, 1, 5, 1), (1, 2, 0, -1), (1, 3, 0, -1)]),      # h_re h_im r_re r_im t_re t_im | c_re c_im
    }

    def __init__(self, kernel_name, flat_param, state1, state2, table_offsets, table_rows, dim, tot_entity, tot_relation,
                 max_pos, neg_rate, sparse=False):
        ns, nd, sites = self.LAYOUTS[kernel_name]
        if len(sites) != len(table_offsets):
            raise L.KgeHipError("staged plan: table list does not match the model")
        dev = flat_param.device
        self.ns, self.nd, self.neg_rate, self.max_pos = ns, nd, int(neg_rate), int(max_pos)
        self.stride = (int(dim) + 3) // 4 * 4
        n_slots = self.max_pos * (ns + nd * self.neg_rate)
        # RotatE rows of more than 512 floats (a bundle's rows over 2 or 4 waves, csrc/kge_score.hip: k_rotate_bundle_staged_split): rows
        # padded to the width the waves cover + one spare slot set behind the used ones -> the kernel's stores carry no predicates
        # (kge_staged_step.stage_spare, round 6)
        self.spare = kernel_name == "rotate" and int(dim) > 512 and os.environ.get("KGE_STAGE_SPARE") != "0"   # (=0: tight rows, A/B)
        if self.spare:
            self.stride = 1024 if int(dim) <= 1024 else 2048
            n_slots += ns + nd
        self.stage = torch.empty(n_slots * self.stride, dtype=torch.float32, device=dev)
        # sparse (SGD / Adagrad with touched-row lists): ONE registration set, count | head back to back so that
        # the train entry point clears it with one memset; the sweep then visits only rows that have a slot.
        # dense: two sets (count, head) alternate between steps and the sweep of a step clears the other one (no memset).
        self.sparse = bool(sparse)
        E = int(tot_entity)
        self.sets = [torch.zeros(2 * E, dtype=torch.int32, device=dev) for _ in range(1 if self.sparse else 2)]
        self.counts = [x[:E] for x in self.sets]
        self.heads = [x[E:2 * E] for x in self.sets]          # overflow chain heads hold pair + 1: zero = empty
        self.dyn_list = torch.zeros(self.max_pos * self.neg_rate, dtype=torch.int32, device=dev) if self.sparse else None
        self.parity = 0
        self.bucket = torch.zeros(tot_entity * STAGED_CAP, dtype=torch.int32, device=dev)
        self.next = torch.zeros(self.max_pos * self.neg_rate, dtype=torch.int32, device=dev)
        self._keep = (flat_param, state1, state2)
        self._cache = {}
        self.n_rel_tables = sum(1 for x in sites if x[0] == 1)
        self.rel_partials = None
        c = self.c = self.cur = L.StagedStep()
        c.param = _dev(flat_param, torch.float32, "param")
        c.state1 = _dev(state1, torch.float32, "state1") if state1 is not None else None
        c.state2 = _dev(state2, torch.float32, "state2") if state2 is not None else None
        for k, ((cls, sa, sb, ds), off, rows) in enumerate(zip(sites, table_offsets, table_rows)):
            c.tables[k].cls, c.tables[k].site_a, c.tables[k].site_b, c.tables[k].dsite = cls, sa, sb, ds
            c.tables[k].flat_off, c.tables[k].rows = int(off), int(rows)
        c.n_tables, c.dim = len(sites), int(dim)
        c.dyn_bucket, c.dyn_next, c.dyn_cap = self.bucket.data_ptr(), self.next.data_ptr(), STAGED_CAP
        c.stage, c.stage_stride = self.stage.data_ptr(), self.stride
        c.stage_spare = 1 if self.spare else 0
        self.dyn_scale = torch.ones(self.max_pos * self.neg_rate, dtype=torch.float32, device=dev) if kernel_name == "rotate" else None
        c.dyn_scale = self.dyn_scale.data_ptr() if self.dyn_scale is not None else None
        c.static_slots, c.dynamic_slots = ns, nd
        c.tot_entity, c.tot_relation = int(tot_entity), int(tot_relation)

    def bind_batch(self, b, index):
        """Point the plan at batch b of a generator.StagedIndex.  The filled descriptor is cached per (batch, parity): the
        eager step is two native calls plus a dictionary lookup, not a dozen pointer validations."""
        key = (b, 0 if self.sparse else self.parity)
        hit = self._cache.get(key)
        if hit is None:
            if self.rel_partials is None and index.max_rel_list > index.LONG_LIST:
                # sized ONCE for the batch with the most chunks: the cached per-batch snapshots below hold this pointer,
                # so the buffer must never be reallocated while the plan lives
                self.rel_partials = torch.empty(max(1, max(index.n_chunks)) * self.n_rel_tables * self.stride,
                                                dtype=torch.float32, device=self.stage.device)
            ent_off, ent_inc, rel_off, rel_inc, n = index.batch(b)
            self.bind(ent_off, ent_inc, rel_off, rel_inc, n, index.chunks(b), index.touched(b) if self.sparse else None)
            snap = L.StagedStep()
            ctypes.memmove(ctypes.byref(snap), ctypes.byref(self.c), ctypes.sizeof(L.StagedStep))
            hit = self._cache[key] = (snap, self._batch, n)
        elif not self.sparse:
            self.parity ^= 1
        self.cur = hit[0]
        return hit[2]

    def bind(self, ent_off, ent_inc, rel_off, rel_inc, n_pos, chunks=None, touched=None):
        if n_pos > self.max_pos:
            raise L.KgeHipError("staged plan: batch larger than the plan")
        self._batch = (ent_off, ent_inc, rel_off, rel_inc, chunks)
        c = self.c
        if chunks is not None:
            chunk_off, chunk_rel, n_chunks = chunks
            need = max(1, n_chunks) * self.n_rel_tables * self.stride
            if self.rel_partials is None or self.rel_partials.numel() < need:
                if self._cache:   # a cached snapshot holds the old pointer: growing now would leave it dangling
                    raise L.KgeHipError("staged plan: relation partial buffer too small for this batch (%d floats needed); "
                                        "bind batches through bind_batch(), which sizes it for the whole index" % need)
                self.rel_partials = torch.empty(need, dtype=torch.float32, device=self.stage.device)
            c.rel_chunk_off, c.chunk_rel = _dev(chunk_off, torch.int32, "rel_chunk_off"), _dev(chunk_rel, torch.int32, "chunk_rel")
            c.rel_partials, c.n_chunks = self.rel_partials.data_ptr(), int(n_chunks)
        else:
            c.rel_chunk_off, c.chunk_rel, c.rel_partials, c.n_chunks = None, None, None, 0
        c.ent_off, c.ent_inc = _dev(ent_off, torch.int32, "ent_off"), _dev(ent_inc, torch.int32, "ent_inc")
        c.rel_off, c.rel_inc = _dev(rel_off, torch.int32, "rel_off"), _dev(rel_inc, torch.int32, "rel_inc")
        c.n_pos, c.n_neg = int(n_pos), int(n_pos) * self.neg_rate
        if self.sparse:
            if touched is None:
                raise L.KgeHipError("staged plan: the sparse sweep needs the batch's touched-row lists")
            t_ent, n_ent, t_rel, n_rel = touched
            self._batch += (t_ent, t_rel)
            c.touched_ent, c.n_touched_ent = _dev(t_ent, torch.int32, "touched_ent"), int(n_ent)
            c.touched_rel, c.n_touched_rel = _dev(t_rel, torch.int32, "touched_rel"), int(n_rel)
            c.dyn_list = self.dyn_list.data_ptr()
            c.dyn_count, c.dyn_head = self.counts[0].data_ptr(), self.heads[0].data_ptr()
            c.dyn_count_next, c.dyn_head_next = None, None
            self.cur = c
            return self
        # this step registers into set `parity`; its optimiser sweep clears the other set for the next step
        q = self.parity
        c.dyn_count, c.dyn_head = self.counts[q].data_ptr(), self.heads[q].data_ptr()
        c.dyn_count_next, c.dyn_head_next = self.counts[q ^ 1].data_ptr(), self.heads[q ^ 1].data_ptr()
        self.parity ^= 1
        self.cur = c
        return self


def train_pairwise_selfadv_sampled_staged(desc, triples, perm, start, n_pos, neg_rate, alpha, bern_prob, slots, seed, offset,
                                          plan, loss_buf):
    """RotatE bundle step with staged (atomic-free) gradient output; follow with optimizer_step_staged(plan)."""
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    L.check(L.load().kge_train_pairwise_selfadv_sampled_staged(
        ctypes.byref(desc), _ids(triples, "triples"), _ids(perm, "perm"), int(start), int(n_pos), int(neg_rate), float(alpha),
        bp, sp, slots.numel() if slots is not None else 0, int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
        ctypes.byref(plan.cur), _dev(loss_buf, torch.float32, "loss"), _stream()), "kge_train_pairwise_selfadv_sampled_staged")


def train_pointwise_logistic_sampled_staged(desc, triples, perm, start, n_pos, neg_rate, bern_prob, slots, seed, offset, lmbda,
                                            reg_type, plan, loss_buf):
    """DistMult / ComplEx bundle step with staged (atomic-free) gradient output; follow with optimizer_step_staged(plan)."""
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    L.check(L.load().kge_train_pointwise_logistic_sampled_staged(
        ctypes.byref(desc), _ids(triples, "triples"), _ids(perm, "perm"), int(start), int(n_pos), int(neg_rate), bp, sp,
        slots.numel() if slots is not None else 0, int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), float(lmbda),
        int(reg_type), ctypes.byref(plan.cur), _dev(loss_buf, torch.float32, "loss"), _stream()),
        "kge_train_pointwise_logistic_sampled_staged")


def optimizer_step_staged(kind, plan, lr, step):
    L.check(L.load().kge_optimizer_step_staged(OPTIMIZER_IDS[kind], ctypes.byref(plan.cur), float(lr), int(step), _stream()),
            "kge_optimizer_step_staged")


def train_pairwise_selfadv(desc, ph, pr, pt, nh, nr, nt, neg_rate, alpha, loss_buf, workspace=None):
    n = ph.numel()
    if nh.numel() != n * neg_rate:
        raise ValueError("negatives must be [B*neg_rate]")
    if workspace is None or workspace.numel() < n * (1 + neg_rate):
        workspace = torch.empty(n * (1 + neg_rate), dtype=torch.float32, device=ph.device)
    L.check(L.load().kge_train_pairwise_selfadv(ctypes.byref(desc), _ids(ph, "ph"), _ids(pr, "pr"), _ids(pt, "pt"),
                                                _ids(nh, "nh"), _ids(nr, "nr"), _ids(nt, "nt"), n, int(neg_rate),
                                                float(alpha), _dev(workspace, torch.float32, "workspace"),
                                                _dev(loss_buf, torch.float32, "loss"), _stream()),
            "kge_train_pairwise_selfadv")
    return workspace


def train_pointwise_logistic(desc, h, r, t, y, lmbda, reg_type, loss_buf, bundle=1):
    n = h.numel()
    L.check(L.load().kge_train_pointwise_logistic(ctypes.byref(desc), _ids(h, "h"), _ids(r, "r"), _ids(t, "t"),
                                                  _ids(y, "y"), n, int(bundle), float(lmbda), int(reg_type),
                                                  _dev(loss_buf, torch.float32, "loss"), _stream()),
            "kge_train_pointwise_logistic")


def l2norm_reg(param, grad, lmbda, loss_buf):
    """NTN.get_reg over one flat parameter buffer: loss += lmbda*||param||_2, grad += lmbda*param/||param||_2."""
    scratch = torch.empty(1, dtype=torch.float32, device=param.device)
    L.check(L.load().kge_l2norm_reg(_dev(param, torch.float32, "param"), _dev(grad, torch.float32, "grad"), param.numel(),
                                    float(lmbda), _dev(scratch, torch.float32, "scratch"),
                                    _dev(loss_buf, torch.float32, "loss"), _stream()), "kge_l2norm_reg")


def optimizer_step(kind, param, grad, state1, state2, lr, step, zero_grad=True, dev_hyper=None):
    p1 = _dev(state1, torch.float32, "state1") if state1 is not None else None
    p2 = _dev(state2, torch.float32, "state2") if state2 is not None else None
    ph = _dev(dev_hyper, torch.float32, "dev_hyper") if dev_hyper is not None else None
    L.check(L.load().kge_optimizer_step(OPTIMIZER_IDS[kind], _dev(param, torch.float32, "param"),
                                        _dev(grad, torch.float32, "grad"), p1, p2, param.numel(), float(lr), int(step),
                                        1 if zero_grad else 0, ph, _stream()), "kge_optimizer_step")


def optimizer_step_rows(kind, param, grad, state1, state2, rows, dim, lr, step, zero_grad=True, normalize=False, dev_hyper=None,
                        touched=None, touched_clear=None):
    """kge_optimizer_step_rows: the dense optimiser with one wave per row of a [rows, dim] table (flat views), optionally storing
    the row renormalised (RESCAL: what the next forward's in-place normalisation would make of it).  touched / touched_clear:
    int32 bitmaps of rows with a gradient (this step's, read; the other parity's, reset)."""
    p1 = _dev(state1, torch.float32, "state1") if state1 is not None else None
    p2 = _dev(state2, torch.float32, "state2") if state2 is not None else None
    ph = _dev(dev_hyper, torch.float32, "dev_hyper") if dev_hyper is not None else None
    L.check(L.load().kge_optimizer_step_rows(OPTIMIZER_IDS[kind], _dev(param, torch.float32, "param"), _dev(grad, torch.float32, "grad"),
                                             p1, p2, int(rows), int(dim), float(lr), int(step), 1 if zero_grad else 0,
                                             1 if normalize else 0, ph,
                                             _dev(touched, torch.int32, "touched") if touched is not None else None,
                                             _dev(touched_clear, torch.int32, "touched_clear") if touched_clear is not None else None,
                                             _stream()), "kge_optimizer_step_rows")


class RescalStage:
    """Buffers of the atomic-free entity gradients of the pairwise RESCAL step (struct kge_rescal_stage): gradient rows staged per
    (pair, side), the pairs' hinge coefficients, and per entity the list of staged rows registered with it (all zero between steps)."""
    CAP = 8

    def __init__(self, tot_entity, n_pairs, dim, device):
        self.n_pairs = int(n_pairs)
        self.gstage = torch.zeros(4 * self.n_pairs * int(dim), dtype=torch.float32, device=device)
        self.dsv = torch.zeros(self.n_pairs, dtype=torch.float32, device=device)
        self.count = torch.zeros(int(tot_entity), dtype=torch.int32, device=device)
        self.bucket = torch.zeros(int(tot_entity) * self.CAP, dtype=torch.int32, device=device)
        self.head = torch.zeros(int(tot_entity), dtype=torch.int32, device=device)
        self.next = torch.zeros(4 * self.n_pairs, dtype=torch.int32, device=device)
        self.c = L.RescalStage(self.gstage.data_ptr(), self.dsv.data_ptr(), self.count.data_ptr(), self.bucket.data_ptr(),
                               self.head.data_ptr(), self.next.data_ptr(), self.CAP)


def rescal_stage_ok(desc, n):
    """kge_rescal_stage_ok: the pairwise RESCAL step of n pairs can stage its entity gradients (slab form, d % 4 == 0, n <= 4096)."""
    return bool(L.load().kge_rescal_stage_ok(ctypes.byref(desc), int(n)))


def rescal_pair_step_staged(desc, ph, pr, pt, nh, nt, margin, loss_buf, touched, stage):
    """kge_rescal_pair_step_staged: the pairwise RESCAL step with every entity gradient row staged instead of added atomically."""
    n = ph.numel()
    if nh.numel() != n or n > stage.n_pairs:
        raise ValueError("rescal_pair_step_staged: %d pairs (stage sized for %d; neg_rate must be 1)" % (n, stage.n_pairs))
    wp, wb, _keep = _workspace(desc, n, ph.device)
    L.check(L.load().kge_rescal_pair_step_staged(ctypes.byref(desc), _ids(ph, "ph"), _ids(pr, "pr"), _ids(pt, "pt"), _ids(nh, "nh"),
                                                 _ids(nt, "nt"), n, float(margin), wp, wb, _dev(loss_buf, torch.float32, "loss"),
                                                 _dev(touched, torch.int32, "touched"), ctypes.byref(stage.c), _stream()),
            "kge_rescal_pair_step_staged")


def optimizer_step_rows_staged(kind, param, grad, state1, state2, rows, dim, lr, step, stage, touched, touched_clear=None, normalize=False,
                               dev_hyper=None):
    """kge_optimizer_step_rows_staged: kge_optimizer_step_rows with the gradient of a touched row summed from the stage."""
    p1 = _dev(state1, torch.float32, "state1") if state1 is not None else None
    p2 = _dev(state2, torch.float32, "state2") if state2 is not None else None
    ph = _dev(dev_hyper, torch.float32, "dev_hyper") if dev_hyper is not None else None
    L.check(L.load().kge_optimizer_step_rows_staged(OPTIMIZER_IDS[kind], _dev(param, torch.float32, "param"), _dev(grad, torch.float32, "grad"),
                                                    p1, p2, int(rows), int(dim), float(lr), int(step), 1 if normalize else 0, ph,
                                                    _dev(touched, torch.int32, "touched"),
                                                    _dev(touched_clear, torch.int32, "touched_clear") if touched_clear is not None else None,
                                                    ctypes.byref(stage.c), _stream()), "kge_optimizer_step_rows_staged")


def rescal_pair_step_ok(desc, n):
    return bool(L.load().kge_rescal_pair_step_ok(ctypes.byref(desc), int(n)))


def rescal_pair_step(desc, ph, pr, pt, nh, nt, margin, loss_buf, touched=None):
    """kge_rescal_pair_step: the pairwise RESCAL step for negatives that keep their positives' relations, one launch per
    (relation, 16 pairs) tile; touched: int32 bitmap [ceil(E / 32)] that receives the entity rows with a gradient."""
    n = ph.numel()
    if nh.numel() != n:
        raise ValueError("pairwise_hinge needs neg_rate == 1 (criterion.py:27 adds [B] to [B*neg_rate])")
    wp, wb, _keep = _workspace(desc, n, ph.device)
    L.check(L.load().kge_rescal_pair_step(ctypes.byref(desc), _ids(ph, "ph"), _ids(pr, "pr"), _ids(pt, "pt"), _ids(nh, "nh"),
                                          _ids(nt, "nt"), n, float(margin), wp, wb, _dev(loss_buf, torch.float32, "loss"),
                                          _dev(touched, torch.int32, "touched") if touched is not None else None, _stream()),
            "kge_rescal_pair_step")


def rescal_normalize_relations(rel, k):
    """The relation-matrix half of Rescal's in-place renormalisation (the entity half rides in kge_optimizer_step_rows)."""
    lib = L.load()
    need = lib.kge_rescal_normalize_scratch_bytes(rel.shape[0], int(k))
    key = (rel.device, need)
    if key not in _rescal_scratch:
        _rescal_scratch[key] = torch.empty(max(1, need // 4), dtype=torch.float32, device=rel.device)
    sc = _rescal_scratch[key]
    L.check(lib.kge_rescal_normalize_ws(None, 0, _dev(rel, torch.float32, "rel"), rel.shape[0], k, sc.data_ptr(), sc.numel() * 4,
                                        _stream()), "kge_rescal_normalize_ws")


def optimizer_step_rownorm_ok(rows, dim):
    return bool(L.load().kge_optimizer_step_rownorm_ok(int(rows), int(dim)))


def optimizer_step_rownorm(kind, param, grad, state1, state2, rows, dim, lr, step, zero_grad=True, advance=None):
    """kge_optimizer_step_rownorm: dense optimiser over a [rows, dim] table of wide rows + the in-place row renormalisation, with the
    rows' sums of squares left by the optimiser launch itself.  advance: None, or (hyper, cursor, next_cursor, next_hyper, batch_stride,
    n_batches, draws_per_batch) -- the step-state transition of optimizer_step_advance folded in."""
    lib = L.load()
    need = lib.kge_rescal_normalize_scratch_bytes(int(rows), int(round(dim ** 0.5))) if int(round(dim ** 0.5)) ** 2 == dim else 4 * int(rows) * ((int(dim) + 4095) // 4096)
    need = max(need, 4 * int(rows) * ((int(dim) + 4095) // 4096))
    key = (param.device, need)
    if key not in _rescal_scratch:
        _rescal_scratch[key] = torch.empty(max(1, need // 4), dtype=torch.float32, device=param.device)
    sc = _rescal_scratch[key]
    p1 = _dev(state1, torch.float32, "state1") if state1 is not None else None
    p2 = _dev(state2, torch.float32, "state2") if state2 is not None else None
    if advance is not None:
        hyper, cursor, next_cursor, next_hyper, batch_stride, n_batches, draws = advance
        adv = (_dev(hyper, torch.float32, "hyper"), _dev(cursor, torch.int64, "cursor"), _dev(next_cursor, torch.int64, "next_cursor"),
               _dev(next_hyper, torch.float32, "next_hyper"), int(batch_stride), int(n_batches), int(draws))
    else:
        adv = (None, None, None, None, 0, 1, 0)
    L.check(lib.kge_optimizer_step_rownorm(OPTIMIZER_IDS[kind], _dev(param, torch.float32, "param"), _dev(grad, torch.float32, "grad"), p1, p2,
                                           int(rows), int(dim), float(lr), int(step), 1 if zero_grad else 0, adv[0], sc.data_ptr(), sc.numel() * 4,
                                           adv[1], adv[2], adv[3], adv[4], adv[5], adv[6], _stream()), "kge_optimizer_step_rownorm")


def optimizer_step_rows_rownorm(kind, param, grad, state1, state2, rows, dim, wparam, wgrad, wstate1, wstate2, wrows, wdim, lr, step,
                                normalize=True, dev_hyper=None, touched=None, touched_clear=None, stage=None, advance=None, zero_grad=True):
    """kge_optimizer_step_rows_rownorm: RESCAL's optimiser step in two launches -- the row-owner sweep of the short-row table with the
    wide-row table's optimiser riding in its first workgroups, then the wide rows' rescale.  Arguments as optimizer_step_rows(_staged) and
    optimizer_step_rownorm; bit-identical to calling those two."""
    lib = L.load()
    need = 4 * int(wrows) * ((int(wdim) + 4095) // 4096)
    key = (param.device, "rider", need)
    if key not in _rescal_scratch:
        _rescal_scratch[key] = torch.empty(max(1, need // 4), dtype=torch.float32, device=param.device)
    sc = _rescal_scratch[key]
    f = lambda t, what: _dev(t, torch.float32, what) if t is not None else None
    if advance is not None:
        hyper, cursor, next_cursor, next_hyper, batch_stride, n_batches, draws = advance
        adv = (_dev(cursor, torch.int64, "cursor"), _dev(next_cursor, torch.int64, "next_cursor"), _dev(next_hyper, torch.float32, "next_hyper"),
               int(batch_stride), int(n_batches), int(draws))
        dev_hyper = hyper
    else:
        adv = (None, None, None, 0, 1, 0)
    L.check(lib.kge_optimizer_step_rows_rownorm(
        OPTIMIZER_IDS[kind], _dev(param, torch.float32, "param"), _dev(grad, torch.float32, "grad"), f(state1, "state1"), f(state2, "state2"),
        int(rows), int(dim), _dev(wparam, torch.float32, "wparam"), _dev(wgrad, torch.float32, "wgrad"), f(wstate1, "wstate1"),
        f(wstate2, "wstate2"), int(wrows), int(wdim), float(lr), int(step), 1 if zero_grad else 0, 1 if normalize else 0, f(dev_hyper, "dev_hyper"),
        _dev(touched, torch.int32, "touched") if touched is not None else None,
        _dev(touched_clear, torch.int32, "touched_clear") if touched_clear is not None else None,
        ctypes.byref(stage.c) if stage is not None else None, sc.data_ptr(), sc.numel() * 4, adv[0], adv[1], adv[2], adv[3], adv[4], adv[5],
        _stream()), "kge_optimizer_step_rows_rownorm")


def optimizer_step_advance(kind, param, grad, state1, state2, lr, hyper, cursor, next_cursor, next_hyper, batch_stride,
                           n_batches, draws_per_batch, zero_grad=True):
    """Dense optimiser sweep of a graph-replayed step + the next step's device-resident state (kge_optimizer_step_advance)."""
    p1 = _dev(state1, torch.float32, "state1") if state1 is not None else None
    p2 = _dev(state2, torch.float32, "state2") if state2 is not None else None
    L.check(L.load().kge_optimizer_step_advance(
        OPTIMIZER_IDS[kind], _dev(param, torch.float32, "param"), _dev(grad, torch.float32, "grad"), p1, p2, param.numel(),
        float(lr), 1 if zero_grad else 0, _dev(hyper, torch.float32, "hyper"), _dev(cursor, torch.int64, "cursor"),
        _dev(next_cursor, torch.int64, "next_cursor"), _dev(next_hyper, torch.float32, "next_hyper"), int(batch_stride),
        int(n_batches), int(draws_per_batch), _stream()), "kge_optimizer_step_advance")


def step_advance(cursor, hyper, batch_stride, n_batches, draws_per_batch, lr):
    """Advance the device-resident step state (int64[8] cursor, float32[4] hyper); first kernel of a captured step."""
    L.check(L.load().kge_step_advance(_dev(cursor, torch.int64, "cursor"), _dev(hyper, torch.float32, "hyper"),
                                      int(batch_stride), int(n_batches), int(draws_per_batch), float(lr), _stream()),
            "kge_step_advance")


def eval_workspace(desc, n, device):
    nbytes = L.load().kge_eval_workspace_bytes(ctypes.byref(desc), int(n))
    if nbytes == 0:
        L.check(-1, "kge_eval_workspace_bytes")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def eval_ranks(desc, triples, tail_off, tail_ids, head_off, head_ids, workspace=None, ties=None):
    """triples int64 [n,3]; CSR filter lists (int64 offsets [n+1], int32 ids) or None.  Returns int32 [4,n]:
    rank_head, rank_tail, filtered_rank_head, filtered_rank_tail (0-based).  ties: optional int32 [2, n] that receives, per head /
    tail sweep, the number of other candidates whose energy equals the true one's (kge_eval_ranks_ties)."""
    n = triples.shape[0]
    if workspace is None:
        workspace = eval_workspace(desc, n, triples.device)
    ranks = torch.empty((4, n), dtype=torch.int32, device=triples.device)
    args = []
    for off, ids in ((tail_off, tail_ids), (head_off, head_ids)):
        if off is None:
            args += [None, None]
        else:
            args += [_dev(off, torch.int64, "csr offsets"), _dev(ids, torch.int32, "csr ids")]
    L.check(L.load().kge_eval_ranks_ties(ctypes.byref(desc), _ids(triples, "triples"), n, *args,
                                         _dev(workspace, torch.uint8, "workspace"), workspace.numel(),
                                         _dev(ranks, torch.int32, "ranks"),
                                         _dev(ties, torch.int32, "ties") if ties is not None else None, _stream()), "kge_eval_ranks")
    return ranks


def eval_ranks_grouped(desc, triples, group_of_triple, group_rel, qblocks, tail_off, tail_ids, head_off, head_ids, ties=None):
    """TransR: triples sorted by relation, several relation groups per call (kge_eval_ranks_grouped).  int32 [4,n]."""
    n, G = triples.shape[0], group_rel.shape[0]
    lib = L.load()
    nbytes = lib.kge_eval_grouped_workspace_bytes(ctypes.byref(desc), int(n), int(G))
    if nbytes == 0:
        L.check(-1, "kge_eval_grouped_workspace_bytes")
    workspace = torch.empty(nbytes, dtype=torch.uint8, device=triples.device)
    ranks = torch.empty((4, n), dtype=torch.int32, device=triples.device)
    args = []
    for off, ids in ((tail_off, tail_ids), (head_off, head_ids)):
        if off is None:
            args += [None, None]
        else:
            args += [_dev(off, torch.int64, "csr offsets"), _dev(ids, torch.int32, "csr ids")]
    L.check(lib.kge_eval_ranks_grouped_ties(ctypes.byref(desc), _ids(triples, "triples"), n,
                                            _dev(group_of_triple, torch.int32, "group_of_triple"), _ids(group_rel, "group_rel"), G,
                                            _dev(qblocks, torch.int32, "qblocks"), qblocks.shape[0], *args,
                                            _dev(workspace, torch.uint8, "workspace"), workspace.numel(),
                                            _dev(ranks, torch.int32, "ranks"),
                                            _dev(ties, torch.int32, "ties") if ties is not None else None, _stream()), "kge_eval_ranks_grouped")
    return ranks


def eval_sweep_scores(desc, triples, workspace=None):
    """float32 [2n, E]: row 2i = energies of (h_i, r_i, e) for all e, row 2i+1 = energies of (e, r_i, t_i)."""
    n = triples.shape[0]
    if workspace is None:
        workspace = eval_workspace(desc, n, triples.device)
    out = torch.empty((2 * n, desc.tot_entity), dtype=torch.float32, device=triples.device)
    L.check(L.load().kge_eval_sweep_scores(ctypes.byref(desc), _ids(triples, "triples"), n,
                                           _dev(workspace, torch.uint8, "workspace"), workspace.numel(),
                                           _dev(out, torch.float32, "scores"), _stream()), "kge_eval_sweep_scores")
    return out


def eval_sweep_scores_side(desc, triples, side, workspace=None):
    """float32 [n, E]: side 0 = energies of (h_i, r_i, e) for all e, side 1 = energies of (e, r_i, t_i); the other side's query
    is never swept (kge_eval_sweep_scores_side).  TransR / NTN compute both sides per call: the full sweep, sliced."""
    if desc.model in (MODEL_IDS["transr"], MODEL_IDS["ntn"]):
        return eval_sweep_scores(desc, triples, workspace)[int(side)::2]
    n = triples.shape[0]
    if workspace is None:
        workspace = eval_workspace(desc, n, triples.device)
    out = torch.empty((n, desc.tot_entity), dtype=torch.float32, device=triples.device)
    L.check(L.load().kge_eval_sweep_scores_side(ctypes.byref(desc), _ids(triples, "triples"), n, int(side),
                                                _dev(workspace, torch.uint8, "workspace"), workspace.numel(),
                                                _dev(out, torch.float32, "scores"), _stream()), "kge_eval_sweep_scores_side")
    return out


def rank_from_scores(scores, truth, off, ids):
    """scores float32 [nq, E] -> (rank, filtered rank) int32 [nq] given the true entity per row and a CSR of knowns."""
    nq, E = scores.shape
    rank = torch.empty(nq, dtype=torch.int32, device=scores.device)
    frank = torch.empty_like(rank)
    po = _dev(off, torch.int64, "csr offsets") if off is not None else None
    pi = _dev(ids, torch.int32, "csr ids") if ids is not None else None
    L.check(L.load().kge_rank_from_scores(_dev(scores, torch.float32, "scores"), nq, E, _ids(truth, "truth"), po, pi,
                                          _dev(rank, torch.int32, "rank"), _dev(frank, torch.int32, "frank"), _stream()),
            "kge_rank_from_scores")
    return rank, frank


def eval_ranks_via_forward(desc, triples, tail_off, tail_ids, head_off, head_ids, chunk=64):
    """The reference's own evaluation algorithm (utils/evaluator.py:254-272) on the device: every candidate triple
    goes through the batch scorer, `chunk` test triples x E candidates per kge_score_forward call, then
    kge_rank_from_scores.  Works for every model; kept as an independent cross-check of the sweep kernels (it is what
    tests compare the NTN MFMA sweep with).  int32 [4, n]."""
    n = triples.shape[0]
    E = desc.tot_entity
    dev = triples.device
    ents = torch.arange(E, dtype=torch.int64, device=dev)
    out = torch.empty((4, n), dtype=torch.int32, device=dev)
    for lo in range(0, n, chunk):
        tr = triples[lo:lo + chunk]
        m = tr.shape[0]
        h, r, t = tr[:, 0:1], tr[:, 1:2], tr[:, 2:3]
        cand = ents.view(1, E).expand(m, E)
        tail_scores = score_forward(desc, h.expand(m, E).reshape(-1).contiguous(), r.expand(m, E).reshape(-1).contiguous(),
                                    cand.reshape(-1).contiguous()).view(m, E)
        head_scores = score_forward(desc, cand.reshape(-1).contiguous(), r.expand(m, E).reshape(-1).contiguous(),
                                    t.expand(m, E).reshape(-1).contiguous()).view(m, E)
        to = (tail_off[lo:lo + m + 1]).contiguous() if tail_off is not None else None
        ho = (head_off[lo:lo + m + 1]).contiguous() if head_off is not None else None
        rt, frt = rank_from_scores(tail_scores, tr[:, 2].contiguous(), to, tail_ids)
        rh, frh = rank_from_scores(head_scores, tr[:, 0].contiguous(), ho, head_ids)
        out[0, lo:lo + m], out[1, lo:lo + m], out[2, lo:lo + m], out[3, lo:lo + m] = rh, rt, frh, frt
    return out


def filter_csr_build(known, queries, tot_entity, tot_relation):
    """Per-query filter lists (tail_off int64 [n+1], tail_ids int32, head_off, head_ids) of `queries` [n, 3] against the set of
    `known` triples [M, 3] (train + valid + test), built on the device: kge_filter_csr_count / _fill.  One host read (the two list
    totals) between the calls."""
    n, M = int(queries.shape[0]), int(known.shape[0])
    dev = queries.device
    t_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    h_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    if n == 0 or M == 0:
        empty = torch.empty(0, dtype=torch.int32, device=dev)
        return t_off, empty, h_off, empty.clone()
    lib = L.load()
    ws = torch.empty(lib.kge_filter_csr_workspace_bytes(M, n), dtype=torch.uint8, device=dev)
    totals = torch.empty(2, dtype=torch.int64, device=dev)
    L.check(lib.kge_filter_csr_count(_ids(known, "known triples"), M, _ids(queries, "queries"), n, int(tot_entity), int(tot_relation),
                                     _dev(ws, torch.uint8, "workspace"), ws.numel(), _dev(t_off, torch.int64, "tail_off"),
                                     _dev(h_off, torch.int64, "head_off"), _dev(totals, torch.int64, "totals"), _stream()),
            "kge_filter_csr_count")
    nt, nh = (int(x) for x in totals.tolist())       # the one host read
    t_ids = torch.empty(max(nt, 1), dtype=torch.int32, device=dev)[:nt]
    h_ids = torch.empty(max(nh, 1), dtype=torch.int32, device=dev)[:nh]
    L.check(lib.kge_filter_csr_fill(_ids(queries, "queries"), n, M, _dev(ws, torch.uint8, "workspace"), _dev(t_off, torch.int64, "tail_off"),
                                    _dev(h_off, torch.int64, "head_off"), ctypes.c_void_p(t_ids.data_ptr()), ctypes.c_void_p(h_ids.data_ptr()),
                                    _stream()), "kge_filter_csr_fill")
    return t_off, t_ids, h_off, h_ids


def triple_set_build(triples):
    """Open-addressing hash set of the train triples (uint64 slots, power of two >= 2n)."""
    n = triples.shape[0]
    if n:  # packed key h:24 | r:16 | t:24 (csrc/kge_sampler_device.h): out-of-range ids would alias silently
        _ids(triples, "triples")
        hi = triples.max(dim=0).values.tolist()
        if triples.min().item() < 0 or max(hi[0], hi[2]) >= 1 << 24 or hi[1] >= 1 << 16:
            raise L.KgeHipError("kge_triple_set_build: ids exceed the packed key (entities < 2^24, relations < 2^16): "
                                "max h/r/t = %s" % hi)
    n_slots = 1 << max(4, math.ceil(math.log2(max(2 * n, 2))))
    slots = torch.empty(n_slots, dtype=torch.int64, device=triples.device)  # uint64 payload
    L.check(L.load().kge_triple_set_build(_ids(triples, "triples"), n, ctypes.c_void_p(slots.data_ptr()), n_slots,
                                          _stream()), "kge_triple_set_build")
    return slots


def corrupt(ph, pr, pt, neg_rate, tot_entity, bern_prob, slots, seed, offset):
    n = ph.numel()
    nh = torch.empty(n * neg_rate, dtype=torch.int64, device=ph.device)
    nr = torch.empty_like(nh)
    nt = torch.empty_like(nh)
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    L.check(L.load().kge_corrupt(_ids(ph, "ph"), _ids(pr, "pr"), _ids(pt, "pt"), n, int(neg_rate), int(tot_entity), bp,
                                 sp, slots.numel() if slots is not None else 0, int(seed) & (2 ** 64 - 1),
                                 int(offset) & (2 ** 64 - 1), _ids(nh, "nh"), _ids(nr, "nr"), _ids(nt, "nt"),
                                 _stream()), "kge_corrupt")
    return nh, nr, nt


def sample_buffer(n_pos, neg_rate, pointwise, device):
    rows = 4 * n_pos * (1 + neg_rate) if pointwise else 3 * n_pos * (1 + neg_rate)
    return torch.empty(rows, dtype=torch.int64, device=device)


def sample_batch(triples, perm, start, n_pos, neg_rate, tot_entity, bern_prob, slots, seed, offset, pointwise=False,
                 out=None, cursor=None):
    """One launch: positives triples[perm[start:start+n_pos]] + their corrupted negatives, in the reference's batch
    layout (pairwise: [ph, pr, pt, nh, nr, nt]; pointwise: [h, r, t, y]).  `out`: optional preallocated
    sample_buffer (static addresses for hipGraph capture); `cursor`: optional device {start, offset} added to the
    host arguments."""
    dev = triples.device
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    pc = _dev(cursor, torch.int64, "cursor") if cursor is not None else None
    if pointwise:
        rows = n_pos * (1 + neg_rate)
        buf = (out if out is not None else torch.empty(4 * rows, dtype=torch.int64, device=dev)).view(4, rows)
        outs = [buf[0], buf[1], buf[2], buf[3]]
        ptrs = [ctypes.c_void_p(o.data_ptr()) for o in outs] + [None, None]
    else:
        nneg = n_pos * neg_rate
        buf = out if out is not None else torch.empty(3 * n_pos + 3 * nneg, dtype=torch.int64, device=dev)
        outs = [buf[0:n_pos], buf[n_pos:2 * n_pos], buf[2 * n_pos:3 * n_pos], buf[3 * n_pos:3 * n_pos + nneg],
                buf[3 * n_pos + nneg:3 * n_pos + 2 * nneg], buf[3 * n_pos + 2 * nneg:]]
        ptrs = [ctypes.c_void_p(o.data_ptr()) for o in outs]
    L.check(L.load().kge_sample_batch(_ids(triples, "triples"), _ids(perm, "perm"), int(start), int(n_pos), int(neg_rate),
                                      int(tot_entity), bp, sp, slots.numel() if slots is not None else 0,
                                      int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), 1 if pointwise else 0,
                                      *ptrs, pc, _stream()), "kge_sample_batch")
    return outs


# ---------------------------------------------------------------------------- owner-computes ("pull") training step
def _i32(t, what):
    return _dev(t, torch.int32, what)


def _ptr_pair(tensors):
    """float* const x[2] argument: a 2-slot pointer array (kept alive by the caller for the duration of the call)."""
    arr = (ctypes.c_void_p * 2)()
    for i in range(2):
        arr[i] = tensors[i].data_ptr() if tensors is not None and tensors[i] is not None else None
    return arr


def row_norms(table, out, hat=None):
    """out[r] = ||table[r]||_2 and hat[r] = table[r] / max(norm, 1e-12), in the lane layout / operation order the pull
    step uses for the rows it writes."""
    L.check(L.load().kge_row_norms(_dev(table, torch.float32, "table"), table.shape[0], table.shape[1],
                                   _dev(out, torch.float32, "norms"),
                                   _dev(hat, torch.float32, "normalised") if hat is not None else None, _stream()),
            "kge_row_norms")


def pull_partial_stride(dim):
    return int(L.load().kge_pull_partial_stride(int(dim)))


def pull_hat_stride(dim):
    """Floats between consecutive rows of the owner-computes step's normalised ("hat") tables."""
    return int(L.load().kge_pull_hat_stride(int(dim)))


def pull_groups_per_block(dim):
    return int(L.load().kge_pull_groups_per_block(int(dim)))


class PullListSet:
    """One kge_pull_lists set: per-step sampler output of the owner-computes steps (count all 0 / head all -1 between steps),
    incl. the ready-made visit descriptors (sdesc: one per static incidence, dbucket: one per bucket entry)."""

    def __init__(self, batch_size, tot_entity, device):
        self.pc = torch.empty(batch_size, dtype=torch.int32, device=device)
        self.next = torch.empty(batch_size, dtype=torch.int32, device=device)
        self.count = torch.zeros(tot_entity, dtype=torch.int32, device=device)
        self.bucket = torch.empty(tot_entity * L.PULL_BUCKET, dtype=torch.int32, device=device)
        self.head = torch.full((tot_entity,), -1, dtype=torch.int32, device=device)
        self.sdesc = torch.empty(3 * batch_size, 4, dtype=torch.int32, device=device)
        self.dbucket = torch.empty(tot_entity * L.PULL_BUCKET, 4, dtype=torch.int32, device=device)
        self.c = L.PullLists(self.pc.data_ptr(), self.count.data_ptr(), self.bucket.data_ptr(), self.head.data_ptr(),
                             self.next.data_ptr(), self.sdesc.data_ptr(), self.dbucket.data_ptr())

    def clear(self):
        self.count.zero_()
        self.head.fill_(-1)


def pull_sample(pairs, inv, tot_entity, bern_prob, slots, seed, offset, lists, cursor=None):
    """Draw the corruption of every pair of a batch (the draws kge_sample_batch makes for the same seed / offset) and
    register each pair with the entity it drew (`lists`: a cleared PullListSet)."""
    bp = _dev(bern_prob, torch.float32, "bern_prob") if bern_prob is not None else None
    sp = ctypes.c_void_p(slots.data_ptr()) if slots is not None else None
    pcur = _dev(cursor, torch.int64, "cursor") if cursor is not None else None
    L.check(L.load().kge_pull_sample(_i32(pairs, "pairs"), _i32(inv, "inv"), pairs.shape[0], int(tot_entity), bp, sp,
                                     slots.numel() if slots is not None else 0, int(seed) & (2 ** 64 - 1),
                                     int(offset) & (2 ** 64 - 1), pcur, ctypes.byref(lists.c), _stream()), "kge_pull_sample")


def pull_lists_explicit(pairs, inv, nh, nt, lists):
    if debug_enabled():   # (the entry point is not told the entity count; the list set is sized by it)
        check_ids(nh, lists.count.numel(), "negative heads")
        check_ids(nt, lists.count.numel(), "negative tails")
    L.check(L.load().kge_pull_lists_explicit(_i32(pairs, "pairs"), _i32(inv, "inv"), _ids(nh, "nh"), _ids(nt, "nt"), pairs.shape[0],
                                             ctypes.byref(lists.c), _stream()), "kge_pull_lists_explicit")


class PullDirection:
    """Scratch of the two-phase ("staged direction") form of the owner-computes step (struct kge_pull_direction): one record and
    two direction codes per pair of the batch."""

    def __init__(self, n_pairs, dim, l1, device):
        cb, rb = ctypes.c_size_t(), ctypes.c_size_t()
        L.check(L.load().kge_pull_direction_bytes(int(dim), 1 if l1 else 0, int(n_pairs), ctypes.byref(cb), ctypes.byref(rb)),
                "kge_pull_direction_bytes")
        self.codes = torch.zeros(max(1, cb.value), dtype=torch.uint8, device=device)
        self.recs = torch.zeros(max(4, rb.value // 4), dtype=torch.float32, device=device)
        self.c = L.PullDirection(self.codes.data_ptr(), self.recs.data_ptr(), int(n_pairs), 0)


def pull_step(desc_in, tables_out, hat_in, hat_out, norm_in, norm_out, state1, state2, pairs, lists, items, inc, partials, multi,
              margin, optimizer, lr, step, loss_buf, reset_lists=True, dev_hyper=None, run_finish=True, sample_next=None,
              dense_skip=None, prepare_only=False, direction=None):
    """One whole training step (scoring, hinge, backward, dense optimiser) without atomics: see csrc/kge_pull.hip.
    desc_in: descriptor over the tables READ; tables_out: [ent, rel] of the other half of the double buffer; hat_in /
    hat_out: the row-normalised copies of both halves.
    sample_next = (next_pairs, next_inv, bern_prob, slots, seed, next_offset, next_lists): the sampler of the next batch rides
    in this launch and fills `next_lists` (a cleared second PullListSet)."""
    to, s1, s2, hi, ho = _ptr_pair(tables_out), _ptr_pair(state1), _ptr_pair(state2), _ptr_pair(hat_in), _ptr_pair(hat_out)
    # run_finish=False (timing only): the owners still write their partial sums, the finishing launch is skipped
    n_multi = multi.shape[0] if (multi is not None and run_finish) else 0
    if sample_next is not None:
        npairs, ninv, bern, slots, seed, noff, nlists = sample_next
        nx = (_i32(npairs, "next_pairs"), _i32(ninv, "next_inv"), npairs.shape[0], _dev(bern, torch.float32, "bern_prob") if bern is not None else None,
              ctypes.c_void_p(slots.data_ptr()) if slots is not None else None, slots.numel() if slots is not None else 0,
              int(seed) & (2 ** 64 - 1), int(noff) & (2 ** 64 - 1), ctypes.byref(nlists.c))
    else:
        nx = (None, None, 0, None, None, 0, 0, 0, None)
    args = (
        ctypes.byref(desc_in), ctypes.addressof(to), ctypes.addressof(hi), ctypes.addressof(ho) if hat_out is not None else None,
        _dev(norm_in, torch.float32, "norm_in"),
        _dev(norm_out, torch.float32, "norm_out") if norm_out is not None else None,
        ctypes.addressof(s1) if state1 is not None else None,
        ctypes.addressof(s2) if state2 is not None else None, _i32(pairs, "pairs"), ctypes.byref(lists.c),
        _i32(items, "items"), items.shape[0], _i32(dense_skip, "dense_skip") if dense_skip is not None else None,
        _i32(inc, "inc"), _dev(partials, torch.float32, "partials"),
        _i32(multi, "multi") if n_multi else None, n_multi, float(margin), OPTIMIZER_IDS[optimizer], float(lr), int(step),
        _dev(dev_hyper, torch.float32, "dev_hyper") if dev_hyper is not None else None, 1 if reset_lists else 0,
        *nx, _dev(loss_buf, torch.float32, "loss"), ctypes.byref(direction.c) if direction is not None else None)
    keep = (desc_in, to, hi, ho, s1, s2, tables_out, hat_in, hat_out, norm_in, norm_out, state1, state2, pairs, lists, items, inc,
            partials, multi, loss_buf, dense_skip, sample_next, direction)
    if prepare_only:   # marshal once, call many times (a data-parallel step runs this launch between two collectives:
        fn = L.load().kge_pull_step                                   # host time per step matters there)
        argl = list(args)
        off_idx = len(argl) - 4      # next_offset: the one argument that differs from epoch to epoch (Philox counters advance)

        def call(next_offset=None):
            if next_offset is not None:
                argl[off_idx] = int(next_offset) & (2 ** 64 - 1)
            L.check(fn(*argl, _stream()), "kge_pull_step")
        call.keep = keep
        return call
    L.check(L.load().kge_pull_step(*args, _stream()), "kge_pull_step")


def pull_index_build(triples, perm, batch_stride, slice_lo, n_pairs, n_batches, tot_entity, tot_relation, segment,
                     groups_per_block, compact):
    """kge_pull_index_build: the incidence index of `n_batches` batches of the permutation, built on the device.  Returns
    (pairs [nb, n, 4], inc [nb, 3n], inv [nb, 3n], items [nb, item_cap, 4], multi [nb, multi_cap, 4], skip [nb, words] or None,
    counts [nb, 4] = {item slots, multi rows, partial slots, listed rows}) -- int32 device tensors at fixed strides."""
    lib = L.load()
    dev = triples.device
    nb, n = int(n_batches), int(n_pairs)
    item_cap, multi_cap, words = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    ws_bytes = ctypes.c_size_t()
    L.check(lib.kge_pull_index_geometry(nb, n, int(tot_entity), int(tot_relation), int(segment), int(groups_per_block),
                                        1 if compact else 0, ctypes.byref(item_cap), ctypes.byref(multi_cap), ctypes.byref(words),
                                        ctypes.byref(ws_bytes)), "kge_pull_index_geometry")
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    pairs, inc, inv = i32(nb, n, 4), i32(nb, 3 * n), i32(nb, 3 * n)
    items, multi = i32(nb, item_cap.value, 4), i32(nb, multi_cap.value, 4)
    skip = i32(nb, words.value) if compact else None
    counts = i32(nb, 4)
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    L.check(lib.kge_pull_index_build(_ids(triples, "triples"), _ids(perm, "perm"), int(batch_stride), int(slice_lo), n, nb,
                                     int(tot_entity), int(tot_relation), int(segment), int(groups_per_block), 1 if compact else 0,
                                     pairs.data_ptr(), inc.data_ptr(), inv.data_ptr(), items.data_ptr(), multi.data_ptr(),
                                     skip.data_ptr() if skip is not None else None, counts.data_ptr(), ws.data_ptr(), ws.numel(),
                                     _stream()), "kge_pull_index_build")
    return pairs, inc, inv, items, multi, skip, counts


def pull_index_bytes(n_batches, n_pairs, tot_entity, tot_relation, segment, groups_per_block, compact):
    """(resident bytes, transient workspace bytes) of kge_pull_index_build's output for these sizes -- the exact strides of
    kge_pull_index_geometry, no device needed."""
    item_cap, multi_cap, words = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    ws_bytes = ctypes.c_size_t()
    L.check(L.load().kge_pull_index_geometry(int(n_batches), int(n_pairs), int(tot_entity), int(tot_relation), int(segment),
                                             int(groups_per_block), 1 if compact else 0, ctypes.byref(item_cap), ctypes.byref(multi_cap),
                                             ctypes.byref(words), ctypes.byref(ws_bytes)), "kge_pull_index_geometry")
    nb, n = int(n_batches), int(n_pairs)
    resident = nb * (n * 16 + 2 * 3 * n * 4 + item_cap.value * 16 + multi_cap.value * 16 + (words.value * 4 if compact else 0) + 16)
    return resident, int(ws_bytes.value)


def pull_list_set_bytes(batch_size, tot_entity):
    """Bytes of ONE PullListSet (two are kept per owner-computes path)."""
    return int(batch_size) * (4 + 4 + 3 * 16) + int(tot_entity) * (4 + 4 + L.PULL_BUCKET * 4 + L.PULL_BUCKET * 16)


class PullPlan:
    """struct kge_pull_plan for a (model, index, state) triple: everything that does not change from step to step,
    marshalled once.  `run` enqueues a whole sequence of steps with ONE foreign call (the per-step work is ~35 us of GPU
    time: a Python-level loop could not keep the queue full)."""

    def __init__(self, model_name, desc_kwargs, tot_entity, tot_relation, tables, hats, norms, state1, state2, lists, index,
                 partials, margin, optimizer, lr, loss_buf, bern, slots, seed, draws_per_batch, direction=None):
        self.keep = (tables, hats, norms, state1, state2, lists, index, partials, loss_buf, bern, slots, direction)   # keep storage alive
        c = L.PullPlanC()
        for half in (0, 1):
            c.model[half] = make_desc(model_name, tables[half], None, tot_entity=tot_entity, tot_relation=tot_relation, **desc_kwargs)
            c.norm[half] = norms[half].data_ptr()
            c.lists[half] = lists[half].c
            for t in (0, 1):
                c.hat[half][t] = hats[half][t].data_ptr()
        for t in (0, 1):
            c.state1[t] = state1[t].data_ptr() if state1 is not None else None
            c.state2[t] = state2[t].data_ptr() if state2 is not None else None
        self.batches = (L.PullBatch * index.n_batches)()
        for b in range(index.n_batches):
            pairs, inc, items, multi = index.batch(b)
            skip = index.skip(b)
            self.batches[b] = L.PullBatch(pairs.data_ptr(), items.data_ptr(), items.shape[0],
                                          skip.data_ptr() if skip is not None else None, inc.data_ptr(), index.inv(b).data_ptr(),
                                          multi.data_ptr() if multi.shape[0] else None, multi.shape[0], pairs.shape[0])
        c.batches = ctypes.cast(self.batches, ctypes.POINTER(L.PullBatch))
        c.n_batches = index.n_batches
        c.partials = partials.data_ptr()
        c.margin, c.optimizer, c.lr = float(margin), OPTIMIZER_IDS[optimizer], float(lr)
        c.bern_prob = bern.data_ptr() if bern is not None else None
        c.slots = slots.data_ptr() if slots is not None else None
        c.n_slots = slots.numel() if slots is not None else 0
        c.seed = int(seed) & (2 ** 64 - 1)
        c.draws_per_batch = int(draws_per_batch)
        c.loss = loss_buf.data_ptr()
        if direction is not None:      # two-phase steps (kge_pull_run fills n_pairs per batch)
            c.direction = direction.c
        self.c = c
        self.fn = L.load().kge_pull_run

    def run(self, first_batch, n_steps, src_half, cur_list, lists_ready, first_opt_step, first_offset, sample_after_last):
        rc = self.fn(ctypes.byref(self.c), int(first_batch), int(n_steps), int(src_half), int(cur_list), 1 if lists_ready else 0,
                     int(first_opt_step), int(first_offset) & (2 ** 64 - 1), int(sample_after_last), _stream())
        if rc:
            L.check(rc, "kge_pull_run")


# ---------------------------------------------------------------------------- TransH / TransD gradients, two-launch owner-computes form
def transx_groups_per_block(dim):
    return int(L.load().kge_transx_groups_per_block(int(dim)))


def transx_partial_stride(dim):
    return int(L.load().kge_transx_partial_stride(int(dim)))


class TransXScratch:
    """Staging rows (5 / 8 gradient rows per pair) and hinge coefficients of kge_transx_grad_step."""

    def __init__(self, model_name, dim, n_pairs, device):
        sb, rb = ctypes.c_size_t(), ctypes.c_size_t()
        L.check(L.load().kge_transx_scratch_bytes(MODEL_IDS[model_name], int(dim), int(n_pairs), ctypes.byref(sb), ctypes.byref(rb)),
                "kge_transx_scratch_bytes")
        self.stage = torch.empty(max(4, sb.value // 4), dtype=torch.float32, device=device)
        self.recs = torch.zeros(max(1, rb.value // 4), dtype=torch.float32, device=device)


def transx_grad_step(desc, pairs, lists, items, listed, inc, partials, multi, margin, scratch, loss_buf, reset_lists=True,
                     sample_next=None, prepare_only=False):
    """kge_transx_grad_step: the TransH / TransD gradients of one batch into desc's dense gradient tables, no float atomics
    (csrc/kge_pullx.hip).  sample_next as pull_step."""
    n_multi = multi.shape[0] if multi is not None else 0
    if sample_next is not None:
        npairs, ninv, bern, slots, seed, noff, nlists = sample_next
        nx = (_i32(npairs, "next_pairs"), _i32(ninv, "next_inv"), npairs.shape[0], _dev(bern, torch.float32, "bern_prob") if bern is not None else None,
              ctypes.c_void_p(slots.data_ptr()) if slots is not None else None, slots.numel() if slots is not None else 0,
              int(seed) & (2 ** 64 - 1), int(noff) & (2 ** 64 - 1), ctypes.byref(nlists.c))
    else:
        nx = (None, None, 0, None, None, 0, 0, 0, None)
    args = [ctypes.byref(desc), _i32(pairs, "pairs"), pairs.shape[0], ctypes.byref(lists.c),
            _i32(items, "items") if items.shape[0] else None, items.shape[0], _i32(listed, "listed") if listed is not None else None,
            _i32(inc, "inc"), _dev(partials, torch.float32, "partials"), _i32(multi, "multi") if n_multi else None, n_multi,
            float(margin), _dev(scratch.stage, torch.float32, "stage"), _dev(scratch.recs, torch.float32, "recs"),
            1 if reset_lists else 0, *nx, _dev(loss_buf, torch.float32, "loss")]
    fn = L.load().kge_transx_grad_step
    if prepare_only:
        off_idx = len(args) - 3      # next_offset: the one argument that changes from epoch to epoch
        keep = (desc, pairs, lists, items, listed, inc, partials, multi, scratch, loss_buf, sample_next)

        def call(next_offset=None):
            if next_offset is not None:
                args[off_idx] = int(next_offset) & (2 ** 64 - 1)
            L.check(fn(*args, _stream()), "kge_transx_grad_step")
        call.keep = keep
        return call
    L.check(fn(*args, _stream()), "kge_transx_grad_step")


class TransXPlan:
    """struct kge_transx_plan: everything of the TransH / TransD step that does not change between steps; `run` enqueues a whole
    sequence of steps (gradients without atomics + flat optimiser) with ONE foreign call."""

    def __init__(self, desc, flat, lists, index, partials, scratch, margin, optimizer, lr, loss_buf, bern, slots, seed, draws_per_batch):
        self.keep = (desc, flat, lists, index, partials, scratch, loss_buf, bern, slots)
        c = L.TransXPlanC()
        ctypes.memmove(ctypes.byref(c.model), ctypes.byref(desc), ctypes.sizeof(L.ModelDesc))
        for half in (0, 1):
            c.lists[half] = lists[half].c
        self.batches = (L.PullBatch * index.n_batches)()
        for b in range(index.n_batches):
            pairs, inc, items, multi = index.batch(b)
            skip = index.skip(b)
            self.batches[b] = L.PullBatch(pairs.data_ptr(), items.data_ptr() if items.shape[0] else None, items.shape[0],
                                          skip.data_ptr() if skip is not None else None, inc.data_ptr(), index.inv(b).data_ptr(),
                                          multi.data_ptr() if multi.shape[0] else None, multi.shape[0], pairs.shape[0])
        c.batches = ctypes.cast(self.batches, ctypes.POINTER(L.PullBatch))
        c.n_batches = index.n_batches
        c.partials, c.stage, c.recs = partials.data_ptr(), scratch.stage.data_ptr(), scratch.recs.data_ptr()
        c.margin = float(margin)
        c.flat_param, c.flat_grad = flat.param.data_ptr(), flat.grad.data_ptr()
        c.flat_state1 = flat.state1.data_ptr() if flat.state1 is not None else None
        c.flat_state2 = flat.state2.data_ptr() if flat.state2 is not None else None
        c.flat_numel = flat.param.numel()
        c.optimizer, c.lr = OPTIMIZER_IDS[optimizer], float(lr)
        c.bern_prob = bern.data_ptr() if bern is not None else None
        c.slots = slots.data_ptr() if slots is not None else None
        c.n_slots = slots.numel() if slots is not None else 0
        c.seed = int(seed) & (2 ** 64 - 1)
        c.draws_per_batch = int(draws_per_batch)
        c.loss = loss_buf.data_ptr()
        self.c = c
        self.fn = L.load().kge_transx_run

    def run(self, first_batch, n_steps, cur_list, lists_ready, first_opt_step, first_offset, sample_after_last):
        rc = self.fn(ctypes.byref(self.c), int(first_batch), int(n_steps), int(cur_list), 1 if lists_ready else 0,
                     int(first_opt_step), int(first_offset) & (2 ** 64 - 1), int(sample_after_last), _stream())
        if rc:
            L.check(rc, "kge_transx_run")


# ---------------------------------------------------------------------------- two-phase owner-computes step (pointwise models)
def own_groups_per_block(model_name, dim):
    return int(L.load().kge_own_groups_per_block(MODEL_IDS[model_name], int(dim)))


def own_partial_stride(model_name, dim):
    return int(L.load().kge_own_partial_stride(MODEL_IDS[model_name], int(dim)))


def own_stage_floats(model_name, dim, n_pairs):
    """Floats of the staged form's buffer for a batch of n_pairs bundles (kge_own_stage_bytes)."""
    return int(L.load().kge_own_stage_bytes(MODEL_IDS[model_name], int(dim), int(n_pairs))) // 4


def _ptr_array(tensors, n=L.KGE_MAX_TABLES):
    arr = (ctypes.c_void_p * n)()
    for i, t in enumerate(tensors or ()):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def own_step(desc, pairs, lists, items, listed, inc, partials, dense, lmbda, reg_type, loss_buf, reset_lists=True, sample_next=None,
             stage=None):
    """Phase 1 (kge_own_step): gradient rows of every touched parameter row into desc.grads, no atomics."""
    if sample_next is not None:
        npairs, ninv, bern, slots, seed, noff, nlists = sample_next
        nx = (_i32(npairs, "next_pairs"), _i32(ninv, "next_inv"), npairs.shape[0], _dev(bern, torch.float32, "bern_prob") if bern is not None else None,
              ctypes.c_void_p(slots.data_ptr()) if slots is not None else None, slots.numel() if slots is not None else 0,
              int(seed) & (2 ** 64 - 1), int(noff) & (2 ** 64 - 1), ctypes.byref(nlists.c))
    else:
        nx = (None, None, 0, None, None, 0, 0, 0, None)
    L.check(L.load().kge_own_step(ctypes.byref(desc), _i32(pairs, "pairs"), pairs.shape[0], ctypes.byref(lists.c), _i32(items, "items"),
                                  items.shape[0], _i32(listed, "listed") if listed is not None else None, _i32(inc, "inc"),
                                  _dev(partials, torch.float32, "partials"), 1 if dense else 0, float(lmbda), int(reg_type),
                                  1 if reset_lists else 0, *nx, _dev(loss_buf, torch.float32, "loss"),
                                  _dev(stage, torch.float32, "stage") if stage is not None else None, _stream()), "kge_own_step")


def own_apply(desc, state1, state2, pairs, lists, items, listed, multi, partials, dense, optimizer, lr, step):
    """Phase 2 (kge_own_apply): the optimiser, in place, on the rows phase 1 produced gradients for."""
    s1, s2 = _ptr_array(state1), _ptr_array(state2)
    n_multi = multi.shape[0] if multi is not None else 0
    L.check(L.load().kge_own_apply(ctypes.byref(desc), ctypes.addressof(s1) if state1 is not None else None,
                                   ctypes.addressof(s2) if state2 is not None else None, _i32(pairs, "pairs"), pairs.shape[0],
                                   ctypes.byref(lists.c), _i32(items, "items"), items.shape[0],
                                   _i32(listed, "listed") if listed is not None else None, _i32(multi, "multi") if n_multi else None,
                                   n_multi, _dev(partials, torch.float32, "partials"), 1 if dense else 0, OPTIMIZER_IDS[optimizer],
                                   float(lr), int(step), _stream()), "kge_own_apply")


class OwnPlan:
    """struct kge_own_plan: everything of the two-phase step that does not change between steps, marshalled once; `run` enqueues
    a whole sequence of steps (two launches each) with ONE foreign call."""

    def __init__(self, desc, state1, state2, lists, index, partials, optimizer, lr, lmbda, reg_type, loss_buf, bern, slots, seed,
                 draws_per_batch, stage=None):
        self.keep = (desc, state1, state2, lists, index, partials, loss_buf, bern, slots, stage)
        c = L.OwnPlanC()
        ctypes.memmove(ctypes.byref(c.model), ctypes.byref(desc), ctypes.sizeof(L.ModelDesc))
        for i, t in enumerate(state1 or ()):
            c.state1[i] = t.data_ptr() if t is not None else None
        for i, t in enumerate(state2 or ()):
            c.state2[i] = t.data_ptr() if t is not None else None
        for half in (0, 1):
            c.lists[half] = lists[half].c
        self.batches = (L.PullBatch * index.n_batches)()
        for b in range(index.n_batches):
            pairs, inc, items, multi = index.batch(b)
            skip = index.skip(b)
            self.batches[b] = L.PullBatch(pairs.data_ptr(), items.data_ptr(), items.shape[0],
                                          skip.data_ptr() if skip is not None else None, inc.data_ptr(), index.inv(b).data_ptr(),
                                          multi.data_ptr() if multi.shape[0] else None, multi.shape[0], pairs.shape[0])
        c.batches = ctypes.cast(self.batches, ctypes.POINTER(L.PullBatch))
        c.n_batches = index.n_batches
        c.partials = partials.data_ptr()
        c.optimizer, c.lr, c.lmbda, c.reg_type = OPTIMIZER_IDS[optimizer], float(lr), float(lmbda), int(reg_type)
        c.bern_prob = bern.data_ptr() if bern is not None else None
        c.slots = slots.data_ptr() if slots is not None else None
        c.n_slots = slots.numel() if slots is not None else 0
        c.seed = int(seed) & (2 ** 64 - 1)
        c.draws_per_batch = int(draws_per_batch)
        c.loss = loss_buf.data_ptr()
        c.stage = stage.data_ptr() if stage is not None else None
        self.c = c
        self.fn = L.load().kge_own_run

    def run(self, first_batch, n_steps, cur_list, lists_ready, first_opt_step, first_offset, sample_after_last):
        rc = self.fn(ctypes.byref(self.c), int(first_batch), int(n_steps), int(cur_list), 1 if lists_ready else 0,
                     int(first_opt_step), int(first_offset) & (2 ** 64 - 1), int(sample_after_last), _stream())
        if rc:
            L.check(rc, "kge_own_run")


# ---------------------------------------------------------------------------- 1-N scoring head (projection models)
def _f32(t, what):
    return _dev(t, torch.float32, what)


def head_1n_forward(x, ent, bias=None, precision="f32"):
    """sigmoid(x @ ent.T + bias): float32 [B, E]  (kge_head_1n_forward; precision="bf16": operands rounded to bfloat16 on their way
    to the matrix cores, fp32 accumulation -- kge_head_1n_forward_bf16)."""
    if precision not in ("f32", "bf16"):
        raise ValueError("head_1n_forward: precision must be 'f32' or 'bf16'")
    B, d = x.shape
    E = ent.shape[0]
    preds = torch.empty((B, E), dtype=torch.float32, device=x.device)
    lib = L.load()
    fn, name = (lib.kge_head_1n_forward, "kge_head_1n_forward") if precision == "f32" else (lib.kge_head_1n_forward_bf16, "kge_head_1n_forward_bf16")
    L.check(fn(_f32(x, "x"), B, d, _f32(ent, "ent"), E, _f32(bias, "bias") if bias is not None else None, _f32(preds, "preds"), _stream()), name)
    return preds


def head_1n_rank(x, ent, bias, truth, off=None, ids=None, return_ties=False, energies=False):
    """(rank, filtered rank) int32 [2, B] of the true entity of every row under sigmoid(x @ ent.T + bias), without the [B, E] tensor
    (kge_head_1n_rank).  truth int64 [B]; off int64 [B+1] / ids int32: CSR of the entities known for each row (None = unfiltered).
    energies=True (tests): the sweep's float32 [B, E] energies -p instead."""
    B, d = x.shape
    E = ent.shape[0]
    dev = x.device
    trip = torch.zeros((B, 3), dtype=torch.int64, device=dev)
    trip[:, 2] = truth
    lib = L.load()
    ws = torch.empty(max(1, lib.kge_head_1n_rank_workspace_bytes(B, d, E, int(bias is not None))), dtype=torch.uint8, device=dev)
    ranks = torch.empty((2, B), dtype=torch.int32, device=dev)
    ties = torch.empty(B, dtype=torch.int32, device=dev) if return_ties else None
    out = torch.empty((B, E), dtype=torch.float32, device=dev) if energies else None
    L.check(lib.kge_head_1n_rank(_f32(x, "x"), B, d, _f32(ent, "ent"), E, _f32(bias.view(-1), "bias") if bias is not None else None,
                                 _ids(trip, "triples"), _dev(off, torch.int64, "csr offsets") if off is not None else None,
                                 _dev(ids, torch.int32, "csr ids") if ids is not None else None,
                                 _dev(ws, torch.uint8, "workspace"), ws.numel(), _dev(ranks, torch.int32, "ranks"),
                                 _dev(ties, torch.int32, "ties") if ties is not None else None,
                                 _f32(out, "energies") if out is not None else None, _stream()), "kge_head_1n_rank")
    if energies:
        return out
    return (ranks, ties) if return_ties else ranks


def head_1n_backward(x, ent, preds, dpreds, need_bias=True, workspace=True):
    """(dx, g_ent, g_bias) of the head given d loss / d preds (kge_head_1n_backward).  workspace=False: the entry point's
    workspace-free form (split-K partial tiles meet in float atomics instead of being added in a fixed order)."""
    B, d = x.shape
    E = ent.shape[0]
    dx = torch.empty_like(x)
    g_ent = torch.zeros_like(ent)
    g_bias = torch.zeros(E, dtype=torch.float32, device=x.device) if need_bias else None
    lib = L.load()
    # (room for the split-K partial tiles: summed in split order instead of with float atomics -- reproducible gradients)
    ws = torch.empty(lib.kge_head_1n_backward_workspace_bytes(), dtype=torch.uint8, device=x.device) if workspace else None
    L.check(lib.kge_head_1n_backward(_f32(x, "x"), B, d, _f32(ent, "ent"), E, _f32(preds, "preds"),
                                     _f32(dpreds, "dpreds"), _f32(dx, "dx"), _f32(g_ent, "g_ent"),
                                     _f32(g_bias, "g_bias") if need_bias else None,
                                     _dev(ws, torch.uint8, "workspace") if workspace else None, ws.numel() if workspace else 0,
                                     _stream()), "kge_head_1n_backward")
    return dx, g_ent, g_bias


def head_1n_bce(x, ent, bias, label_off, label_ids, label_smoothing, loss_buf, g_ent, g_bias=None):
    """One direction of Criterion.multi_class_bce fused with the head and its backward (kge_head_1n_bce).
    label_off int64 [B+1], label_ids int32 [n_pos]; label_smoothing None = off.  Adds to loss_buf / g_ent / g_bias,
    returns dx [B, d]."""
    B, d = x.shape
    E = ent.shape[0]
    n_pos = int(label_ids.numel())
    lib = L.load()
    ws = torch.empty(lib.kge_head_1n_bce_workspace_bytes(B, E, n_pos), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x)
    L.check(lib.kge_head_1n_bce(_f32(x, "x"), B, d, _f32(ent, "ent"), E, _f32(bias, "bias") if bias is not None else None,
                                _dev(label_off, torch.int64, "label_off"),
                                _dev(label_ids, torch.int32, "label_ids") if n_pos else None, n_pos,
                                -1.0 if label_smoothing is None else float(label_smoothing),
                                _dev(ws, torch.uint8, "workspace"), ws.numel(), _f32(loss_buf, "loss"), _f32(dx, "dx"),
                                _f32(g_ent, "g_ent"), _f32(g_bias, "g_bias") if g_bias is not None else None, _stream()),
            "kge_head_1n_bce")
    return dx
