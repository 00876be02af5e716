"""The exact lazy form of the dense optimisers (csrc/kge_opt.hip: kge_lazy_catchup / kge_optimizer_step_rows_lazy / kge_lazy_flush)
against the dense sweep it replaces (kge_optimizer_step_rows over every row every step, torch.optim semantics on dense nn.Embedding
gradients, utils/trainer.py:112-131 + models/Domain.py:8-13): BIT-identical tables and optimiser state, with and without RESCAL's
per-step row renormalisation (pairwise.py:843-844), and through the Trainer's RESCAL epoch (eager and hipGraph-replayed)."""
import numpy as np
import pytest
import torch

import kge_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hip_util
    return hip_util


def _bitmap_np(rows, n_rows, dev):
    w = np.zeros((n_rows + 31) // 32, dtype=np.uint32)
    r = np.unique(np.asarray(rows, dtype=np.int64))
    np.bitwise_or.at(w, r >> 5, (np.uint32(1) << (r & 31).astype(np.uint32)))
    return torch.from_numpy(w.view(np.int32)).to(dev)


@pytest.mark.parametrize("kind", ["adam", "rms", "adagrad", "sgd"])
@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("rows,dim,touch", [(300, 100, 12), (97, 8, 3), (1000, 200, 40), (64, 400, 5), (130, 1024, 4)])
def test_lazy_rows_equal_the_dense_sweep_bit_for_bit(hip, kind, normalize, rows, dim, touch):
    from pykg2vec_amd import kernels as K
    dev = "cuda"
    rng = np.random.default_rng(rows * 7 + dim + len(kind) + int(normalize))
    T, lr = 60, 0.01
    p0 = rng.normal(size=(rows, dim)).astype(np.float32) * 0.1
    if normalize:
        p0 /= np.linalg.norm(p0, axis=1, keepdims=True)
    mk = lambda a: torch.from_numpy(a.copy()).to(dev)
    state = lambda: (torch.zeros(rows, dim, device=dev) if kind != "sgd" else None, torch.zeros(rows, dim, device=dev) if kind == "adam" else None)
    pd, (md, vd) = mk(p0), state()
    pl, (ml, vl) = mk(p0), state()
    gd, gl = torch.zeros(rows, dim, device=dev), torch.zeros(rows, dim, device=dev)
    lazy = K.LazyRows(rows, lr, dev)
    bm = [torch.zeros((rows + 31) // 32, dtype=torch.int32, device=dev) for _ in range(2)]
    never = set(range(rows))
    for t in range(1, T + 1):
        # a few rows get a gradient; some rows are read without getting one (a pair inside its margin); some steps touch nothing
        S = rng.choice(rows, size=touch, replace=False) if t % 7 else np.zeros(0, np.int64)
        read_only = rng.choice(rows, size=2, replace=False)
        never -= set(S.tolist())
        g = rng.normal(size=(len(S), dim)).astype(np.float32)
        if len(S):
            gd[mk(S)] = mk(g)
            gl[mk(S)] = mk(g)
        K.optimizer_step_rows(kind, pd.view(-1), gd.view(-1), md.view(-1) if md is not None else None, vd.view(-1) if vd is not None else None,
                              rows, dim, lr, t, zero_grad=True, normalize=normalize)
        ids = np.concatenate([S, read_only, read_only[:1]])           # duplicates: a row named twice is replayed once
        pad = np.resize(ids, ((len(ids) + 1) // 2) * 2).reshape(2, -1)  # two id lists of equal length
        K.lazy_catchup(kind, pl.view(-1), ml.view(-1) if ml is not None else None, vl.view(-1) if vl is not None else None, rows, dim, lr,
                       lazy, t, [mk(pad[0]), mk(pad[1])], normalize=normalize)
        par = t & 1
        bm[par].copy_(_bitmap_np(S, rows, dev)) if len(S) else bm[par].zero_()
        if t % 11 == 0 and len(S):     # a stale bit of an earlier step with this parity: a row that is NOT current must stay behind
            extra = np.setdiff1d(np.arange(rows), ids)[:1]
            bm[par] |= _bitmap_np(extra, rows, dev)
        K.optimizer_step_rows_lazy(kind, pl.view(-1), gl.view(-1), ml.view(-1) if ml is not None else None, vl.view(-1) if vl is not None else None,
                                   rows, dim, lr, t, lazy, bm[par], bm[1 - par], normalize=normalize)
        assert torch.equal(gl, torch.zeros_like(gl))
        if len(S):   # the rows of this step are current in both forms
            assert torch.equal(pl[mk(S)], pd[mk(S)]), t
    assert len(never) > 0 or rows <= touch * T   # (some rows were never touched: the all-zero-state fast path ran)
    behind = int((lazy.last.cpu().numpy() < T).sum())
    assert behind > 0
    K.lazy_flush(kind, pl.view(-1), ml.view(-1) if ml is not None else None, vl.view(-1) if vl is not None else None, rows, dim, lr, lazy, T,
                 normalize=normalize, normalize_last=True)
    assert torch.equal(pl, pd)
    if md is not None:
        assert torch.equal(ml, md)
    if vd is not None:
        assert torch.equal(vl, vd)
    assert bool((lazy.last == T).all())


def test_flush_leaves_the_last_step_unnormalised_when_asked(hip):
    """The epoch's last optimiser step is not followed by a renormalisation (that belongs to the next forward): dense = T - 1
    normalising steps + one plain step."""
    from pykg2vec_amd import kernels as K
    dev, rows, dim, T, lr = "cuda", 50, 32, 9, 0.05
    rng = np.random.default_rng(1)
    p0 = rng.normal(size=(rows, dim)).astype(np.float32)
    m0 = rng.normal(size=(rows, dim)).astype(np.float32) * 0.01
    v0 = np.abs(rng.normal(size=(rows, dim))).astype(np.float32) * 1e-4
    mk = lambda a: torch.from_numpy(a.copy()).to(dev)
    pd, md, vd, g = mk(p0), mk(m0), mk(v0), torch.zeros(rows, dim, device=dev)
    for t in range(1, T + 1):
        K.optimizer_step_rows("adam", pd.view(-1), g.view(-1), md.view(-1), vd.view(-1), rows, dim, lr, t, normalize=t < T)
    pl, ml, vl = mk(p0), mk(m0), mk(v0)
    lazy = K.LazyRows(rows, lr, dev)
    K.lazy_flush("adam", pl.view(-1), ml.view(-1), vl.view(-1), rows, dim, lr, lazy, T, normalize=True, normalize_last=False)
    assert torch.equal(pl, pd) and torch.equal(ml, md) and torch.equal(vl, vd)


def test_step_beyond_the_hyper_table_is_refused_and_ensure_grows_it(hip):
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd._lib import KgeHipError
    lazy = K.LazyRows(8, 0.01, "cuda")
    p, m, v = (torch.zeros(8, 4, device="cuda") for _ in range(3))
    with pytest.raises(KgeHipError, match="outside the hyper table"):
        K.lazy_flush("adam", p.view(-1), m.view(-1), v.view(-1), 8, 4, 0.01, lazy, lazy.cap + 5)
    assert lazy.ensure(lazy.cap + 5) and not lazy.ensure(10)
    K.lazy_flush("adam", p.view(-1), m.view(-1), v.view(-1), 8, 4, 0.01, lazy, 70000)
    # the table IS torch's bias correction in double, rounded once
    h = lazy.hyper.cpu().numpy()
    for t in (1, 2, 1000, 65535):
        assert h[t, 0] == np.float32(0.01 / (1.0 - 0.9 ** t)) and h[t, 1] == np.float32(np.sqrt(1.0 - 0.999 ** t))


def _rescal_trainer(hip, optimizer, lazy, use_graph, E=3000, R=5, k=20, B=64, n_train=64 * 12):
    import os
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(11)
    hp = dict(hidden_size=k, margin=1.0)
    P = ko.init_params("rescal", rng, tot_entity=E, tot_relation=R, hidden_size=k)
    trip = np.stack([rng.integers(E, size=n_train + 40), rng.integers(R, size=n_train + 40), rng.integers(E, size=n_train + 40)], 1)
    cfg = hip.make_config(E, R, hp, trip[:n_train], trip[n_train:n_train + 20], trip[n_train + 20:], optimizer=optimizer, lr=0.01, batch_size=B)
    cfg.seed = 3
    m = hip.model_from_params("rescal", P, hp, E, R)
    tr = Trainer(m, cfg, use_graph=use_graph)
    tr.switches["rescal_fused"] = True          # (the fused row-owner optimiser is the default only for tables >= 32 MB)
    tr.switches["lazy_opt"] = lazy
    tr.build_model()
    tr.generator = tr._new_generator()
    return m, tr


@pytest.mark.parametrize("optimizer", ["adam", "rms", "adagrad", "sgd"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_rescal_epochs_with_the_lazy_optimiser_equal_the_dense_epochs_bit_for_bit(hip, optimizer, use_graph):
    """Three epochs of 12 steps (>= 50 optimiser steps would be the verdict's wording: 5 epochs below) through
    Trainer.train_model_epoch, lazy against dense: tables, optimiser state and epoch losses identical."""
    runs = []
    for lazy in (False, True):
        m, tr = _rescal_trainer(hip, optimizer, lazy, use_graph)
        losses = [tr.train_model_epoch(e) for e in range(5)]
        torch.cuda.synchronize()
        assert (getattr(tr, "_lazy", None) is not None) == lazy
        if lazy:
            assert bool((tr._lazy.last == tr.flat.step).all()) and tr.flat.step == 60
        runs.append((losses, tr.flat.param.clone(), None if tr.flat.state1 is None else tr.flat.state1.clone(),
                     None if tr.flat.state2 is None else tr.flat.state2.clone()))
    (la, pa, s1a, s2a), (lb, pb, s1b, s2b) = runs
    assert la == lb
    assert torch.equal(pa, pb)
    assert s1a is None or torch.equal(s1a, s1b)
    assert s2a is None or torch.equal(s2a, s2b)


def test_lazy_rescal_epoch_then_evaluation_sees_current_tables(hip):
    """The flush at the end of an epoch: an Evaluator run right after a lazy epoch ranks with the same tables as after a dense one."""
    from pykg2vec_amd.evaluator import Evaluator
    out = []
    for lazy in (False, True):
        m, tr = _rescal_trainer(hip, "adam", lazy, False)
        tr.train_model_epoch(0)
        ev = Evaluator(m, tr.config)
        out.append(ev.rank_all(tr.config.knowledge_graph.cache["triplets_test"], 16).clone())
    assert torch.equal(out[0], out[1])
