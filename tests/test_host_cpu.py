"""CPU-only checks of the boundary and host logic: the C-ABI library loads and exports every symbol
include/kge_hip.h declares (no compute calls), the drop-in classes keep the reference's construction /
naming contract (pinned by the key names inside tests/golden, which come from the live reference), the filter
CSR / metric code agrees with the oracle, and the product path refuses to run without the GPU."""
import ctypes
import types
import os
import re

import numpy as np
import pytest
import torch

import kge_oracle as ko
from golden_util import CASES, Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from pykg2vec_amd import _lib
    header = open(os.path.join(ROOT, "include", "kge_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(kge_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().kge_abi_version() == _lib.ABI_VERSION


def test_model_desc_struct_matches_header_layout():
    from pykg2vec_amd import _lib
    # int32 model, uint32 flags, 2x int64, 2x int32, 2x float, 12+12 pointers (KGE_MAX_TABLES)
    assert _lib.KGE_MAX_TABLES == 12
    assert ctypes.sizeof(_lib.ModelDesc) == 4 + 4 + 8 + 8 + 4 + 4 + 4 + 4 + 8 * 24
    assert _lib.ModelDesc.tables.offset == 40 and _lib.ModelDesc.grads.offset == 40 + 8 * 12


def test_argument_validation_without_gpu():
    from pykg2vec_amd import _lib
    lib = _lib.load()
    d = _lib.ModelDesc()
    d.model = 99
    assert lib.kge_score_forward(ctypes.byref(d), None, None, None, 4, None, None, 0, None) != 0
    assert b"unknown model" in lib.kge_last_error()
    d.model, d.dim, d.tot_entity, d.tot_relation = _lib.TRANSE, 8, 10, 3
    assert lib.kge_score_forward(ctypes.byref(d), None, None, None, 4, None, None, 0, None) != 0
    assert b"table 0 is null" in lib.kge_last_error()
    assert lib.kge_optimizer_step(0, None, None, None, None, 16, 0.1, 1, 1, None, None) != 0
    assert lib.kge_triple_set_build(None, 5, None, 7, None) != 0  # 7 is not a power of two


@pytest.mark.parametrize("name", CASES)
def test_dropin_classes_keep_reference_contract(name):
    import pykg2vec_amd as pa
    from pykg2vec_amd.common import TrainingStrategy
    c = Case(name)
    cls = pa.import_model(c.model)
    kw = dict(c.hp, tot_entity=c.E, tot_relation=c.R, device="cpu", batch_size=c.B, tot_train_triples=len(c.train),
              knowledge_graph=types.SimpleNamespace(read_cache_data=lambda key: c.train))
    m = cls(**kw)
    if c.model == "transm":  # the fixed per-relation weights are derived from the train split exactly as the reference does
        assert np.array_equal(m.theta.numpy(), c.z["theta"])
    ref_keys = sorted(k[len("init."):] for k in c.z.files if k.startswith("init."))
    assert sorted(m.state_dict().keys()) == ref_keys                       # reference checkpoint key names
    for k in ref_keys:
        assert tuple(m.state_dict()[k].shape) == c.z["init." + k].shape
    assert [p.weight.shape for p in m.parameter_list] == [c.z["init.%s.weight" % n].shape for n in ko.PARAM_NAMES[c.model]]
    assert all(hasattr(p, "name") for p in m.parameter_list)
    assert m.model_name == c.model
    want = TrainingStrategy.POINTWISE_BASED if c.pointwise else TrainingStrategy.PAIRWISE_BASED
    assert m.training_strategy == want
    assert callable(m.loss) and callable(m.get_reg) and callable(m.embed)
    m.load_state_dict({k: torch.from_numpy(c.z["init." + k]) for k in ref_keys})  # reference weights load as-is
    first = next(iter(kw))
    bad = dict(kw)
    bad.pop("tot_entity")
    with pytest.raises(Exception, match="hyperparameter tot_entity not found!"):
        cls(**bad)


def test_embed_matches_reference_tuple_shapes():
    import pykg2vec_amd.pairwise as pw
    import pykg2vec_amd.pointwise as pt
    h = torch.tensor([1, 2]); r = torch.tensor([0, 1]); t = torch.tensor([3, 4])
    assert len(pw.TransE(tot_entity=9, tot_relation=3, hidden_size=8, l1_flag=True).embed(h, r, t)) == 3
    assert len(pw.RotatE(tot_entity=9, tot_relation=3, hidden_size=8, margin=6.0).embed(h, r, t)) == 6
    assert len(pt.Complex(tot_entity=9, tot_relation=3, hidden_size=8, lmbda=0.1).embed(h, r, t)) == 6
    a = pw.TransH(tot_entity=9, tot_relation=3, hidden_size=8, l1_flag=True)
    eh, er, et = a.embed(h, r, t)
    w = torch.nn.functional.normalize(a.w(r), dim=-1)
    assert torch.allclose((eh * w).sum(-1), torch.zeros(2), atol=1e-6)  # projected onto the hyperplane


def test_losses_match_oracle():
    from pykg2vec_amd.criterion import Criterion
    rng = np.random.default_rng(0)
    pos = rng.normal(size=12).astype(np.float32); neg = rng.normal(size=36).astype(np.float32)
    got = Criterion.pariwise_logistic(torch.from_numpy(pos), torch.from_numpy(neg), 3, 0.7).item()
    assert np.isclose(got, ko.pairwise_logistic_selfadv(pos, neg, 3, 0.7)[0], rtol=1e-5)
    assert np.isclose(Criterion.pairwise_hinge(torch.from_numpy(pos), torch.from_numpy(neg[:12]), 0.8).item(),
                      ko.pairwise_hinge(pos, neg[:12], 0.8)[0], rtol=1e-5)
    y = np.where(rng.random(12) > 0.5, 1, -1)
    assert np.isclose(Criterion.pointwise_logistic(torch.from_numpy(pos), torch.from_numpy(y).float()).item(),
                      ko.pointwise_logistic(pos, y)[0], rtol=1e-5)


def test_filter_csr_and_metrics_match_oracle():
    import types
    from pykg2vec_amd.evaluator import MetricCalculator, build_filter_csr
    c = Case("transe_l1")
    hr_t, tr_h = c.filters()
    t_off, t_ids, h_off, h_ids = build_filter_csr(c.test, hr_t, tr_h)
    for i, (h, r, t) in enumerate(c.test):
        assert set(t_ids[t_off[i]:t_off[i + 1]]) == hr_t[(int(h), int(r))]
        assert set(h_ids[h_off[i]:h_off[i + 1]]) == tr_h[(int(t), int(r))]
    kg = types.SimpleNamespace(read_cache_data=lambda k: {"hr_t": hr_t, "tr_h": tr_h}[k])
    cfg = types.SimpleNamespace(knowledge_graph=kg, hits=[1, 3, 5, 10], tot_entity=c.E, tot_relation=c.R)
    mc = MetricCalculator(cfg)
    ranks = np.stack([c.z["eval.rank_head"], c.z["eval.rank_tail"], c.z["eval.frank_head"], c.z["eval.frank_tail"]])
    mc.append_ranks(ranks, 0)
    mc.settle()
    for k in ("mr", "fmr", "mrr", "fmrr"):
        assert np.isclose(mc.get_curr_scores()[k], c.z["eval." + k], rtol=1e-6)   # reference's own settled metrics
    for hit in (1, 3, 5, 10):
        assert np.isclose(mc.hit[(0, hit)], c.z["eval.hit%d" % hit]) and np.isclose(mc.fhit[(0, hit)], c.z["eval.fhit%d" % hit])


def test_product_path_fails_loudly_without_gpu():
    import pykg2vec_amd.pairwise as pw
    from pykg2vec_amd._lib import KgeHipError
    m = pw.TransE(tot_entity=10, tot_relation=3, hidden_size=8, l1_flag=True)
    with pytest.raises(KgeHipError, match="no CPU fallback"):
        m(torch.tensor([1]), torch.tensor([1]), torch.tensor([1]))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pykg2vec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "kge_oracle" not in text and "oracle_backend" not in text and "/root/reference" not in text, f


def test_stand_alone_trainer_stops_early_checkpoints_the_best_and_keeps_reference_checkpoint_format(tmp_path):
    """train_model honours config.patience with the reference's rule (a default PatienceStopper, any object with should_stop can be
    injected instead), keeps the best weights under config.save_model, and save_model / load_model use the reference's file
    names, key names and default directory (utils/trainer.py:86-87,199-219,388-419); a wrong checkpoint raises."""
    import oracle_backend
    import hip_util
    from pykg2vec_amd.trainer import Trainer, PatienceStopper
    c = Case("transe_l1")
    cfg = hip_util.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test, optimizer="sgd", lr=0.05, batch_size=64, device="cpu")
    cfg.epochs, cfg.test_step, cfg.test_num = 5, 1, 4
    cfg.tot_train_triples = 64

    class StopAtSecond:
        calls = 0

        def should_stop(self, metrics):
            assert "fmr" in metrics
            self.calls += 1
            return self.calls >= 2

    m = hip_util.model_from_case(c, device="cpu")
    tr = Trainer(m, cfg, backend=oracle_backend)
    tr.build_model()
    assert isinstance(tr.early_stopper, PatienceStopper) and tr.early_stopper.patience == cfg.patience
    tr.early_stopper = StopAtSecond()
    assert tr.train_model() == 1 and tr.early_stopper.calls == 2
    tr.save_model(tmp_path)
    assert os.path.exists(os.path.join(str(tmp_path), Trainer.TRAINED_MODEL_CONFIG_NAME))
    state = torch.load(os.path.join(str(tmp_path), Trainer.TRAINED_MODEL_FILE_NAME))
    assert set(state) == {"ent_embeddings.weight", "rel_embeddings.weight"}
    m2 = hip_util.model_from_case(c, device="cpu")
    tr2 = Trainer(m2, cfg, backend=oracle_backend)
    tr2.build_model()
    tr2.load_model(tmp_path)
    for k, v in state.items():
        assert torch.equal(dict(m2.named_parameters())[k].detach(), v)
    # strict: a checkpoint with a missing key, an unexpected key or a wrong shape raises
    bad_dir = tmp_path / "bad"
    bad_dir.mkdir()
    for bad in ({"ent_embeddings.weight": state["ent_embeddings.weight"]},
                dict(state, extra=torch.zeros(1)),
                dict(state, **{"rel_embeddings.weight": state["rel_embeddings.weight"][:, :1].clone()})):
        torch.save(bad, str(bad_dir / Trainer.TRAINED_MODEL_FILE_NAME))
        with pytest.raises(ValueError, match="does not match the model"):
            tr2.load_model(bad_dir)
    with pytest.raises(ValueError, match="Cannot load model"):
        tr2.load_model(tmp_path / "nowhere")
    # default directory = config.path_tmp / model_name, and the best weights are kept while training under config.save_model
    cfg.path_tmp, cfg.save_model, cfg.epochs = tmp_path / "tmp", True, 2
    m3 = hip_util.model_from_case(c, device="cpu")
    tr3 = Trainer(m3, cfg, backend=oracle_backend)
    tr3.build_model()
    tr3.train_model()
    assert tr3.best_metric is not None
    assert os.path.exists(os.path.join(str(cfg.path_tmp), "transe", Trainer.TRAINED_MODEL_FILE_NAME))
    tr3.load_model()


def test_patience_rule_equals_the_reference_early_stopper():
    """PatienceStopper.should_stop == the reference's EarlyStopper.should_stop on the same metric sequences, for every monitor and
    for zero / positive / negative patience (utils/trainer.py:22-68).  Needs the reference tree (build container only)."""
    import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present")
    ref_shim.install()
    from pykg2vec.utils.trainer import EarlyStopper
    from pykg2vec.common import Monitor as RefMonitor
    from pykg2vec_amd.trainer import PatienceStopper
    from pykg2vec_amd.common import Monitor
    rng = np.random.default_rng(0)
    for mon in ("mr", "fmr", "mrr", "fmrr"):
        for patience in (-1, 0, 1, 3):
            for trial in range(20):
                ours, ref = PatienceStopper(patience, Monitor(mon)), EarlyStopper(patience, RefMonitor(mon))
                seq = rng.integers(0, 4, size=25).astype(float)          # plenty of equal and worse neighbours
                for v in seq:
                    metrics = {"mr": v, "fmr": v, "mrr": v, "fmrr": v}
                    assert ours.should_stop(metrics) == ref.should_stop(metrics), (mon, patience, trial)


def test_relation_property_matches_reference_rule():
    from pykg2vec_amd.generator import relation_property
    c = Case("transe_l1")
    assert np.allclose(relation_property(c.train, c.R), ko.bern_probability(c.train, c.R))
    sparse = c.train[c.train[:, 1] != 3]
    got = relation_property(sparse, c.R)
    assert got[3] == 0.0 and np.allclose(got, ko.bern_probability(sparse, c.R))


def test_vectorised_filter_csr_equals_dict_of_sets_form():
    from pykg2vec_amd.evaluator import build_filter_csr, build_filter_csr_from_triples
    c = Case("transe_l1")
    hr_t, tr_h = c.filters()
    allt = np.concatenate([c.train, c.valid, c.test, c.train[:50]])  # duplicates must not matter
    q = np.concatenate([c.test, np.asarray([[c.E - 1, c.R - 1, 0]])])  # last query may have no known entity at all
    a = build_filter_csr(q, hr_t, tr_h)
    b = build_filter_csr_from_triples(q, allt, c.R)
    for side in (0, 2):
        assert np.array_equal(a[side], b[side])
        for i in range(len(q)):
            assert set(a[side + 1][a[side][i]:a[side][i + 1]]) == set(b[side + 1][b[side][i]:b[side][i + 1]])


def test_new_entry_points_validate_arguments_without_gpu():
    from pykg2vec_amd import _lib
    lib = _lib.load()
    d = _lib.ModelDesc()
    d.model, d.dim, d.rel_dim, d.tot_entity, d.tot_relation = _lib.TRANSE, 8, 8, 10, 3
    fake = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    for i in range(2):
        d.tables[i] = 16
    assert lib.kge_eval_ranks_grouped(ctypes.byref(d), fake, 4, fake, fake, 1, fake, 1, None, None, None, None, fake, 1 << 20,
                                      fake, None) != 0
    assert b"TransR / TransH / TransD only" in lib.kge_last_error()
    assert lib.kge_head_1n_forward(None, 4, 8, None, 10, None, None, None) != 0
    assert lib.kge_head_1n_bce_workspace_bytes(0, 10, 0) == 0
    assert lib.kge_optimizer_step_advance(1, fake, fake, fake, fake, 16, 0.1, 1, fake, fake, fake, fake, 4, 2, 4, None) != 0
    assert b"different set" in lib.kge_last_error()  # next-step state must not alias the current one
    d.model, d.dim, d.rel_dim = _lib.TRANSR, 200, 50
    d.tables[2] = 16
    assert lib.kge_score_forward(ctypes.byref(d), fake, fake, fake, 4, fake, fake, 1 << 20, None) != 0
    assert b"exceed the LDS-resident tile kernel" in lib.kge_last_error()


def test_pull_plan_struct_layout_matches_the_library():
    """struct kge_pull_plan is filled by the Python binding and read by kge_pull_run: the layouts must agree."""
    from pykg2vec_amd import _lib
    lib = _lib.load()
    assert lib.kge_pull_plan_bytes() == ctypes.sizeof(_lib.PullPlanC)
    assert ctypes.sizeof(_lib.PullBatch) == 72 and ctypes.sizeof(_lib.PullLists) == 56
    assert lib.kge_own_plan_bytes() == ctypes.sizeof(_lib.OwnPlanC)
    assert lib.kge_transx_plan_bytes() == ctypes.sizeof(_lib.TransXPlanC)
    assert ctypes.sizeof(_lib.PullDirection) == 32
    assert lib.kge_transx_groups_per_block(100) == 8 and lib.kge_transx_partial_stride(100) == 256 and lib.kge_transx_groups_per_block(600) == 0
    assert lib.kge_staged_step_bytes() == ctypes.sizeof(_lib.StagedStep)
    assert lib.kge_pull_partial_stride(100) == 128 and lib.kge_pull_partial_stride(102) == 0   # rows move as float4
    assert lib.kge_pull_groups_per_block(100) in (8, 16) and lib.kge_pull_groups_per_block(1000) == 8
    assert lib.kge_pull_run(None, 0, 1, 0, 0, 0, 1, 0, 0, None) != 0 and b"kge_pull_run" in lib.kge_last_error()


def test_staged_index_is_the_incidence_csr_of_every_batch():
    """generator.StagedIndex (the static half of the staged optimiser sweep's inverse index): per batch, entity e lists the
    positives it heads / tails as positive << 1 | side in ascending order, relation r the positives it labels; chunk offsets
    cut a relation's list into REL_CHUNK-slot pieces; touched lists are the sorted unique ids."""
    from pykg2vec_amd.generator import StagedIndex
    rng = np.random.default_rng(3)
    E, R = 17, 3
    batches = [np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1) for n in (40, 40, 9)]
    ix = StagedIndex(batches, E, R, "cpu")
    for b, pos in enumerate(batches):
        ent_off, ent_inc, rel_off, rel_inc, n = ix.batch(b)
        assert n == len(pos) and int(ent_off[-1]) == 2 * n and int(rel_off[-1]) == n
        for e in range(E):
            want = sorted([2 * i for i in range(n) if pos[i, 0] == e] + [2 * i + 1 for i in range(n) if pos[i, 2] == e])
            assert ent_inc[int(ent_off[e]):int(ent_off[e + 1])].tolist() == want
        for r in range(R):
            assert rel_inc[int(rel_off[r]):int(rel_off[r + 1])].tolist() == [i for i in range(n) if pos[i, 1] == r]
        t_ent, n_ent, t_rel, n_rel = ix.touched(b)
        assert t_ent[:n_ent].tolist() == sorted(set(pos[:, 0]) | set(pos[:, 2])) and t_rel[:n_rel].tolist() == sorted(set(pos[:, 1]))
    assert ix.max_rel_list < StagedIndex.LONG_LIST and ix.chunks(0) is None
    big = [np.stack([rng.integers(E, size=300), np.zeros(300, np.int64), rng.integers(E, size=300)], 1)]
    ix2 = StagedIndex(big, E, R, "cpu")
    chunk_off, chunk_rel, n_chunks = ix2.chunks(0)
    c = StagedIndex.REL_CHUNK
    assert n_chunks == (300 + c - 1) // c and chunk_off.tolist() == [0, n_chunks, n_chunks, n_chunks]
    assert chunk_rel[:n_chunks].tolist() == [0] * n_chunks


def test_pull_index_of_a_data_parallel_rank_covers_its_slice_of_every_batch():
    """At world_size N the owner-computes gradient step of rank r works on pairs [r B/N, (r+1) B/N) of every batch -- the
    slice Generator._next_range hands that rank."""
    import hip_util
    import oracle_backend
    from pykg2vec_amd.generator import Generator
    rng = np.random.default_rng(5)
    E, R, n_train, B = 30, 4, 100, 32
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    hp = dict(hidden_size=8, l1_flag=True, margin=1.0)
    cfg = hip_util.make_config(E, R, hp, train, train[:2], train[:2], batch_size=B, device="cpu")
    m = hip_util.model_from_params("transe", {}, hp, E, R, device="cpu")
    for rank in (0, 1):
        gen = Generator(m, cfg, rank=rank, world_size=2, backend=oracle_backend)
        gen.K = types.SimpleNamespace(pull_groups_per_block=lambda d: 8)   # (the geometry query is the library's)
        idx = gen.pull_index()
        assert idx.batch_size == B // 2 and idx.n_batches == n_train // B
        gen.start_one_epoch(idx.n_batches)
        perm = gen._perm_np
        for b in range(idx.n_batches):
            gen.K = oracle_backend
            start, n, _ = gen._next_range()
            assert n == B // 2 and start == b * B + rank * (B // 2)
            pairs = idx.batch(b)[0].numpy()
            assert np.array_equal(pairs[:, :3], train[perm[start:start + n]])


@pytest.mark.parametrize("E,R,B,seg", [(53, 7, 32, 8), (53, 7, 32, 1), (500, 20, 40, 8), (14951, 1345, 128, 8)])
def test_compact_pull_index_visits_every_row_exactly_once(E, R, B, seg):
    """generator.build_pull_batch(compact=True): explicit items for the rows with an incidence, a bitmap of those rows, and the
    kernel's implicit enumeration (every row whose bit is clear, after the explicit items) together cover each parameter row
    once; the explicit part is the full index restricted to the touched rows."""
    from pykg2vec_amd.generator import build_pull_batch
    rng = np.random.default_rng(E + B)
    pos = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
    full = build_pull_batch(pos, E, R, seg, 8)
    pairs, inc, items, multi, nglob, words = build_pull_batch(pos, E, R, seg, 8, compact=True)
    assert np.array_equal(pairs, full[0]) and np.array_equal(inc, full[1]) and np.array_equal(multi, full[3]) and nglob == full[4]
    nrows = E + R
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:nrows].astype(bool)
    touched = np.zeros(nrows, bool)
    touched[pos[:, 0]] = True; touched[pos[:, 2]] = True; touched[E + pos[:, 1]] = True
    assert np.array_equal(bits, touched)
    live = items[items[:, 0] >= 0]
    assert set(live[:, 0]) == set(np.flatnonzero(touched)) and len(items) % 8 == 0
    # the same (row, first, end, info) items as the full index has for those rows
    f_live = full[2][full[2][:, 0] >= 0]
    want = {tuple(x) for x in f_live if touched[x[0]]}
    assert {tuple(x) for x in live} == want
    implicit = [g for g in range(nrows) if not bits[g]]
    rows_once = sorted(set(live[:, 0])) + implicit
    assert sorted(rows_once) == list(range(nrows))


@pytest.mark.parametrize("E,R,B,seg,gpb", [(53, 7, 32, 8, 8), (53, 7, 64, 2, 8), (40, 3, 200, 1, 8), (300, 5, 1000, 8, 4),
                                            (14951, 1345, 4096, 8, 8)])
def test_pull_index_structure(E, R, B, seg, gpb):
    """generator.build_pull_batch invariants the kernel relies on: every parameter row is covered by work items whose
    incidence ranges tile its sorted (pair, role) list exactly; items of at most `seg` incidences; rows of 2..gpb items sit in
    consecutive slots of ONE workgroup in segment order (kind 3); longer rows are kind 1 (first) / 2 (rest) with consecutive
    partial slots listed in `multi`; slots are padded per workgroup with row -1."""
    from pykg2vec_amd.generator import build_pull_batch
    rng = np.random.default_rng(E * 7 + B)
    pos = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
    pairs, inc, items, multi, nglob = build_pull_batch(pos, E, R, seg, gpb)
    assert np.array_equal(pairs[:, :3], pos) and len(inc) == 3 * B and len(items) % gpb == 0
    nrows = E + R
    # the incidence list: sorted by (row, pair, role); role 0 = head, 1 = tail, 2 = relation
    rows_of = np.where((inc & 3) == 0, pos[inc >> 2, 0], np.where((inc & 3) == 1, pos[inc >> 2, 2], E + pos[inc >> 2, 1]))
    assert np.all(np.diff(rows_of) >= 0)
    for g in np.unique(rows_of):
        seg_vals = inc[rows_of == g]
        assert np.all(np.diff(seg_vals) > 0)
    counts = np.bincount(rows_of, minlength=nrows)
    start = np.cumsum(counts) - counts
    covered = {}
    for slot, (g, lo, hi, info) in enumerate(items):
        if g < 0:
            continue
        assert 0 <= hi - lo <= seg and (hi > lo or counts[g] == 0)
        covered.setdefault(int(g), []).append((int(lo), int(hi), int(info), slot))
    assert sorted(covered) == list(range(nrows))
    n_partial = 0
    for g, its in covered.items():
        its.sort()
        assert its[0][0] == start[g] and its[-1][1] == start[g] + counts[g]
        assert all(a[1] == b[0] for a, b in zip(its, its[1:]))
        kinds = [x[2] & 3 for x in its]
        if len(its) == 1:
            assert kinds == [0]
        elif len(its) <= gpb:
            assert kinds == [3] * len(its)
            slots = [x[3] for x in its]
            assert slots == list(range(slots[0], slots[0] + len(its))) and slots[0] // gpb == slots[-1] // gpb
            assert [(x[2] >> 2) & 15 for x in its] == list(range(len(its))) and all((x[2] >> 6) == len(its) for x in its)
        else:
            assert kinds == [1] + [2] * (len(its) - 1)
            ps = [x[2] >> 2 for x in its]
            assert ps == list(range(ps[0], ps[0] + len(its)))
            row = multi[multi[:, 0] == g]
            assert len(row) == 1 and row[0, 1] == ps[0] and row[0, 2] == len(its)
            n_partial += len(its)
    assert n_partial == nglob


def test_copies_of_a_model_get_their_own_op_handle():
    """torch.ops.kge.score finds its model through an integer handle (pykg2vec_amd/ops.py): deepcopy / pickle must not share it."""
    import copy
    import pickle
    import pykg2vec_amd.pointwise as pt
    from pykg2vec_amd import ops
    m = pt.Complex(tot_entity=10, tot_relation=3, hidden_size=8, lmbda=0.1)
    clones = [copy.deepcopy(m), pickle.loads(pickle.dumps(m))]
    keys = {m._kge_op_key} | {c._kge_op_key for c in clones}
    assert len(keys) == 3 and all(ops._model(c._kge_op_key) is c for c in clones + [m])
    assert str(torch.ops.kge.score.default._schema).startswith("kge::score(SymInt key, Tensor h, Tensor r, Tensor t, Tensor[] weights)")
