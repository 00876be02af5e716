"""Failing loudly on bad ids (the reference raises IndexError from nn.Embedding, models/Domain.py:8-13; here the debug mode of
include/kge_hip.h: kge_set_debug / KGE_DEBUG_IDS) and the device-built filter lists of the rank sweep (kge_filter_csr_*) against
the dict-of-sets form the reference keeps (data/kgcontroller.py:410-428)."""
import numpy as np
import pytest
import torch

import kge_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hip_util
    return hip_util


@pytest.fixture()
def debug_ids():
    from pykg2vec_amd import kernels as K
    K.set_debug(True)
    yield K
    K.set_debug(False)


def _transe(hip, E=50, R=7, d=20, optimizer="sgd", batch=16):
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(3)
    hp = dict(hidden_size=d, l1_flag=True, margin=1.0)
    P = ko.init_params("transe", rng, tot_entity=E, tot_relation=R, hidden_size=d)
    m = hip.model_from_params("transe", P, hp, E, R)
    trip = np.stack([rng.integers(E, size=96), rng.integers(R, size=96), rng.integers(E, size=96)], 1)
    cfg = hip.make_config(E, R, hp, trip[:64], trip[64:80], trip[80:], optimizer=optimizer, batch_size=batch)
    tr = Trainer(m, cfg, use_graph=False)
    tr.build_model()
    return m, tr, cfg, trip


def test_debug_mode_is_off_by_default_and_switchable(hip):
    from pykg2vec_amd import kernels as K
    assert not K.debug_enabled()
    K.set_debug(True)
    assert K.debug_enabled()
    K.set_debug(False)
    assert not K.debug_enabled()


@pytest.mark.parametrize("column", ["head", "relation", "tail"])
def test_forward_raises_on_an_id_beyond_its_table(hip, debug_ids, column):
    from pykg2vec_amd._lib import KgeHipError
    m, tr, cfg, trip = _transe(hip)
    h, r, t = (trip[:8, k].copy() for k in range(3))
    {"head": h, "relation": r, "tail": t}[column][5] = {"head": 50, "relation": 7, "tail": 10 ** 9}[column]
    with pytest.raises(KgeHipError, match=r"%s id out of range at position 5" % column):
        m(hip.dev(h), hip.dev(r), hip.dev(t))
    # in range: the same call goes through with the mode on
    out = m(hip.dev(trip[:8, 0]), hip.dev(trip[:8, 1]), hip.dev(trip[:8, 2]))
    assert out.shape == (8,)


def test_negative_id_raises(hip, debug_ids):
    from pykg2vec_amd._lib import KgeHipError
    m, tr, cfg, trip = _transe(hip)
    h = trip[:8, 0].copy()
    h[0] = -1
    with pytest.raises(KgeHipError, match="head id out of range at position 0: -1"):
        m(hip.dev(h), hip.dev(trip[:8, 1]), hip.dev(trip[:8, 2]))


def test_train_step_entry_points_raise(hip, debug_ids):
    from pykg2vec_amd._lib import KgeHipError
    m, tr, cfg, trip = _transe(hip)
    pos = trip[:16]
    neg = pos.copy()
    neg[3, 2] = 50       # == tot_entity
    args = [hip.dev(x) for x in (pos[:, 0], pos[:, 1], pos[:, 2], neg[:, 0], neg[:, 1], neg[:, 2])]
    with pytest.raises(KgeHipError, match=r"kge_train_pairwise_hinge \(negatives\): tail id out of range at position 3"):
        tr.train_step_pairwise(*args)
    # pointwise entry point
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(1)
    hp = dict(hidden_size=8, lmbda=0.1)
    P = ko.init_params("distmult", rng, tot_entity=20, tot_relation=3, hidden_size=8)
    dm = hip.model_from_params("distmult", P, hp, 20, 3)
    tri = np.stack([rng.integers(20, size=40), rng.integers(3, size=40), rng.integers(20, size=40)], 1)
    c2 = hip.make_config(20, 3, hp, tri[:30], tri[30:35], tri[35:], optimizer="adagrad", batch_size=8)
    t2 = Trainer(dm, c2, use_graph=False)
    t2.build_model()
    h, r, t = hip.dev(tri[:8, 0]), hip.dev(np.full(8, 3)), hip.dev(tri[:8, 2])
    y = hip.dev(np.tile([1, -1], 4))
    with pytest.raises(KgeHipError, match="kge_train_pointwise_logistic: relation id out of range at position 0: 3"):
        t2.train_step_pointwise(h, r, t, y)


def test_sampler_index_and_hash_set_raise_on_a_bad_train_triple(hip, debug_ids):
    """The device sampler / owner-computes index take their ids from the train split: a bad triple there is reported when the
    hash set or the index is built, before any step runs."""
    from pykg2vec_amd._lib import KgeHipError
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(5)
    E, R, d = 40, 4, 16
    hp = dict(hidden_size=d, l1_flag=True, margin=1.0)
    P = ko.init_params("transe", rng, tot_entity=E, tot_relation=R, hidden_size=d)
    m = hip.model_from_params("transe", P, hp, E, R)
    trip = np.stack([rng.integers(E, size=64), rng.integers(R, size=64), rng.integers(E, size=64)], 1)
    bad = trip.copy()
    bad[17, 0] = E + 3
    cfg = hip.make_config(E, R, hp, bad, trip[:8], trip[8:16], optimizer="adam", batch_size=16)
    tr = Trainer(m, cfg, use_graph=False)
    tr.build_model()
    with pytest.raises(KgeHipError, match="id out of range"):
        tr.generator = tr._new_generator()
        tr.generator.start_one_epoch(4)
        tr.step_next_batches(4)
        torch.cuda.synchronize()


def test_eval_raises_on_a_bad_query(hip, debug_ids):
    from pykg2vec_amd._lib import KgeHipError
    from pykg2vec_amd.evaluator import Evaluator
    m, tr, cfg, trip = _transe(hip)
    ev = Evaluator(m, cfg)
    q = trip[80:].copy()
    q[2, 1] = 7
    with pytest.raises(KgeHipError, match="relation id out of range at position 2: 7"):
        ev.rank_all(q, len(q))
    assert ev.rank_all(trip[80:], 16).shape == (4, 16)


def test_check_skipped_during_graph_capture_and_off_when_disabled(hip):
    """With the mode off the entry points do not scan (and therefore do not synchronise); an in-range batch gives the same scores
    with the mode on and off."""
    from pykg2vec_amd import kernels as K
    m, tr, cfg, trip = _transe(hip)
    args = [hip.dev(trip[:8, k]) for k in range(3)]
    a = m(*args).detach().cpu().numpy()
    K.set_debug(True)
    try:
        b = m(*args).detach().cpu().numpy()
        g = torch.cuda.CUDAGraph()
        desc = m.make_desc()
        with torch.cuda.graph(g):   # a scan would synchronise the capturing stream (illegal): it must be skipped
            out = K.score_forward(desc, *args)
        g.replay()
        torch.cuda.synchronize()
        c = out.cpu().numpy()
    finally:
        K.set_debug(False)
    assert np.array_equal(a, b) and np.array_equal(a, c)


# ---------------------------------------------------------------------------- table shapes (make_desc)
def test_quate_with_more_relations_than_entities_is_refused(hip):
    """QuatE's relation tables carry tot_entity rows (models/pointwise.py:622-631) and are looked up by relation id: with
    tot_relation > tot_entity the reference raises IndexError from nn.Embedding; the descriptor is refused here."""
    from pykg2vec_amd._lib import KgeHipError
    import pykg2vec_amd as pa
    kw = dict(tot_entity=5, tot_relation=9, hidden_size=8, lmbda=0.1, device="cuda", batch_size=4, tot_train_triples=10)
    m = pa.import_model("quate")(**kw).to("cuda")
    with pytest.raises(KgeHipError, match="indexed by relation ids up to 8"):
        m.make_desc()
    ok = pa.import_model("quate")(**dict(kw, tot_entity=9, tot_relation=5)).to("cuda")
    ok.make_desc()


def test_table_with_too_few_rows_or_wrong_width_is_refused(hip):
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd._lib import KgeHipError
    ent = torch.zeros(10, 8, device="cuda")
    rel = torch.zeros(3, 8, device="cuda")
    K.make_desc("transe", [ent, rel], None, tot_entity=10, tot_relation=3, dim=8)
    with pytest.raises(KgeHipError, match="table 0 has 10 rows but is indexed by entity ids up to 10"):
        K.make_desc("transe", [ent, rel], None, tot_entity=11, tot_relation=3, dim=8)
    with pytest.raises(KgeHipError, match=r"table 1 must be \[rows, 16\]"):
        K.make_desc("transe", [torch.zeros(10, 16, device="cuda"), rel], None, tot_entity=10, tot_relation=3, dim=16)
    with pytest.raises(KgeHipError, match="table 1 must be"):
        K.make_desc("rescal", [ent, torch.zeros(3, 60, device="cuda")], None, tot_entity=10, tot_relation=3, dim=8)


# ---------------------------------------------------------------------------- filter lists built on the device
def _dict_csr(queries, known):
    hr_t, tr_h = {}, {}
    for h, r, t in known:
        hr_t.setdefault((int(h), int(r)), set()).add(int(t))
        tr_h.setdefault((int(t), int(r)), set()).add(int(h))
    return ([sorted(hr_t.get((int(h), int(r)), ())) for h, r, t in queries],
            [sorted(tr_h.get((int(t), int(r)), ())) for h, r, t in queries])


@pytest.mark.parametrize("E,R,M,n,seed", [(30, 3, 500, 64, 0), (14951, 1345, 60000, 3000, 1), (7, 1, 2000, 50, 2), (200, 5, 1, 9, 3),
                                          (1000, 11, 2047, 300, 4), (1000, 11, 2049, 1, 5)])
def test_device_filter_csr_equals_the_dict_of_sets(hip, E, R, M, n, seed):
    from pykg2vec_amd import kernels as K
    rng = np.random.default_rng(seed)
    known = np.stack([rng.integers(E, size=M), rng.integers(R, size=M), rng.integers(E, size=M)], 1).astype(np.int64)
    known = np.concatenate([known, known[: M // 3]])          # duplicates across splits must not repeat an id
    take = rng.integers(len(known), size=n)
    queries = known[take].copy()
    queries[::5, 0] = rng.integers(E, size=len(queries[::5]))  # some queries whose (h, r) has no known tail
    t_off, t_ids, h_off, h_ids = K.filter_csr_build(hip.dev(known), hip.dev(queries), E, R)
    t_off, t_ids, h_off, h_ids = (x.cpu().numpy() for x in (t_off, t_ids, h_off, h_ids))
    want_t, want_h = _dict_csr(queries, known)
    assert t_off[0] == 0 and h_off[0] == 0 and t_off[-1] == len(t_ids) and h_off[-1] == len(h_ids)
    for i in range(n):
        assert list(t_ids[t_off[i]:t_off[i + 1]]) == want_t[i], i
        assert list(h_ids[h_off[i]:h_off[i + 1]]) == want_h[i], i


def test_largest_packable_ids_survive_the_key(hip):
    from pykg2vec_amd import kernels as K
    E, R = 1 << 24, 1 << 16
    known = np.array([[E - 1, R - 1, E - 1], [E - 1, R - 1, 0], [0, 0, E - 1], [E - 1, R - 1, E - 1]], dtype=np.int64)
    q = np.array([[E - 1, R - 1, E - 1], [0, 0, E - 1]], dtype=np.int64)
    t_off, t_ids, h_off, h_ids = (x.cpu().numpy() for x in K.filter_csr_build(hip.dev(known), hip.dev(q), E, R))
    assert list(t_ids[t_off[0]:t_off[1]]) == [0, E - 1] and list(h_ids[h_off[0]:h_off[1]]) == [E - 1]
    assert list(t_ids[t_off[1]:t_off[2]]) == [E - 1] and list(h_ids[h_off[1]:h_off[2]]) == [0]


def test_evaluator_ranks_are_the_same_from_triples_and_from_dicts(hip):
    from pykg2vec_amd.evaluator import Evaluator
    m, tr, cfg, trip = _transe(hip, E=200, R=5)
    a, b = Evaluator(m, cfg), Evaluator(m, cfg)
    a.FILTER_SOURCE, b.FILTER_SOURCE = "triples", "dicts"
    ra, rb = a.rank_all(trip[80:], 16), b.rank_all(trip[80:], 16)
    assert torch.equal(ra, rb)
    assert a.setup_stats["csr_source"] == "triples" and b.setup_stats["csr_source"] == "dicts"
    assert Evaluator(m, cfg)._filter_source() == "triples"       # arrays in the cache: the device build is the default


def test_evaluator_cache_is_keyed_by_content_not_identity(hip):
    """Two different arrays of the same length (what a freed-and-reallocated id would look like) get their own filter lists."""
    from pykg2vec_amd.evaluator import Evaluator
    m, tr, cfg, trip = _transe(hip, E=200, R=5)
    ev = Evaluator(m, cfg)
    q1 = trip[80:96].copy()
    r1 = ev.rank_all(q1, 16).clone()
    q1[:] = trip[64:80]                       # same object, new content
    r2 = ev.rank_all(q1, 16)
    fresh = Evaluator(m, cfg).rank_all(trip[64:80].copy(), 16)
    assert torch.equal(r2, fresh) and not torch.equal(r1, r2)
