"""Edge cases of the HIP path against the oracle: empty and ragged batches, every row-length boundary of the register
geometries, entity counts around the 64-candidate sweep tile, extreme ids, duplicated rows (gradient collisions),
all-zero embedding rows (the eps branch of F.normalize), oversize hidden size (must fail loudly)."""
import numpy as np
import pytest
import torch

import kge_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hip_util
    return hip_util


def _setup(hip, model, hp, E, R, rng, optimizer="sgd"):
    from pykg2vec_amd.trainer import Trainer
    shape_kw = {k: v for k, v in hp.items() if k in ("hidden_size", "ent_hidden_size", "rel_hidden_size")}
    if model == "rotate":
        shape_kw["margin"] = hp["margin"]
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, **shape_kw)
    m = hip.model_from_params(model, P, hp, E, R)
    trip = np.stack([rng.integers(E, size=64), rng.integers(R, size=64), rng.integers(E, size=64)], 1)
    cfg = hip.make_config(E, R, dict(hp, margin=hp.get("margin", 1.0)), trip, trip[:4], trip, optimizer=optimizer)
    tr = Trainer(m, cfg, use_graph=False)
    tr.build_model()
    return P, m, tr, cfg, trip


@pytest.mark.parametrize("d", [1, 2, 31, 32, 33, 64, 65, 127, 128, 129, 256, 257, 512, 513, 1024])
def test_every_row_length_boundary(hip, d):
    rng = np.random.default_rng(d)
    hp = dict(hidden_size=d, l1_flag=bool(d % 2), margin=1.0)
    E, R, n = 40, 5, 37  # 37: ragged against 8 groups per workgroup
    P, m, tr, cfg, _ = _setup(hip, "transe", hp, E, R, rng)
    ids = [rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)]
    neg = [ids[0].copy(), ids[1].copy(), rng.integers(E, size=n)]
    batch = (ids[0], ids[1], ids[2], neg[0], neg[1], neg[2])
    loss_ref, G, sc, _ = ko.train_step_grads("transe", P, batch, **hp)
    with torch.no_grad():
        got = m(*[hip.dev(a) for a in ids]).cpu().numpy()
    assert np.allclose(got, sc[0], atol=2e-5, rtol=2e-5)
    loss = tr.train_step_pairwise(*[hip.dev(a) for a in batch]).item()
    assert np.isclose(loss, loss_ref, rtol=5e-5, atol=5e-5)
    for name, g in zip(("ent_embeddings", "rel_embeddings"), tr.flat.grad_views):
        assert np.allclose(g.cpu().numpy(), G[name], atol=5e-5, rtol=2e-4), name


def test_hidden_size_beyond_register_kernels_fails_loudly(hip):
    from pykg2vec_amd._lib import KgeHipError
    rng = np.random.default_rng(0)
    hp = dict(hidden_size=2049, l1_flag=True, margin=1.0)
    P, m, tr, cfg, trip = _setup(hip, "transe", hp, 8, 2, rng)
    with pytest.raises(KgeHipError, match="exceeds"):
        m(hip.dev(trip[:4, 0] % 8), hip.dev(trip[:4, 1] % 2), hip.dev(trip[:4, 2] % 8))


@pytest.mark.parametrize("model,hp", [("transe", dict(hidden_size=20, l1_flag=True, margin=1.0)),
                                      ("distmult", dict(hidden_size=20, lmbda=0.01)),
                                      ("rotate", dict(hidden_size=20, margin=6.0, neg_rate=2, alpha=1.0)),
                                      ("rescal", dict(hidden_size=8, margin=1.0)),
                                      ("ntn", dict(ent_hidden_size=8, rel_hidden_size=4, lmbda=0.1, margin=1.0))])
def test_empty_inputs_are_no_ops(hip, model, hp):
    from pykg2vec_amd import kernels as K
    rng = np.random.default_rng(1)
    P, m, tr, cfg, trip = _setup(hip, model, hp, 30, 4, rng)
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        assert m(e, e, e).numel() == 0
    before = tr.flat.grad.clone()
    if model in ko.POINTWISE:
        tr.train_step_pointwise(e, e, e, e)
    else:
        tr.train_step_pairwise(e, e, e, e, e, e)
    if model == "ntn":  # NTN.get_reg (pairwise.py:962-963) does not depend on the batch: it still applies
        assert np.isclose(K.read_loss(tr.loss_buf).item(), ko.ntn_reg(P, hp["lmbda"])[0], rtol=1e-5)
    else:
        assert torch.equal(before, tr.flat.grad) and K.read_loss(tr.loss_buf).item() == 0.0
    from pykg2vec_amd.evaluator import Evaluator
    assert Evaluator(m, cfg).rank_all(trip, 0).shape == (4, 0)


@pytest.mark.parametrize("E", [1, 2, 63, 64, 65, 127, 128, 129, 200])
def test_entity_counts_around_the_sweep_tile(hip, E):
    from pykg2vec_amd.evaluator import Evaluator
    rng = np.random.default_rng(E)
    hp = dict(hidden_size=12, l1_flag=True, margin=1.0)
    R = 3
    P, m, tr, cfg, _ = _setup(hip, "transe", hp, E, R, rng)
    test = np.stack([rng.integers(E, size=9), rng.integers(R, size=9), rng.integers(E, size=9)], 1)
    test[0] = (0, 0, E - 1)          # extreme ids
    test[1] = (E - 1, R - 1, 0)
    hr_t, tr_h = ko.build_filters(test)
    # the known triples are these nine: the splits (what the device builds its filter lists from) and the dicts must say the same
    cfg.knowledge_graph.cache.update(hr_t=hr_t, tr_h=tr_h, triplets_train=test[:5], triplets_valid=test[5:7], triplets_test=test[7:])
    got = Evaluator(m, cfg).rank_all(test, 9).cpu().numpy()
    _, rk = ko.evaluate("transe", P, test, hr_t, tr_h, l1_flag=True)
    ref = np.stack([rk["head"], rk["tail"], rk["fhead"], rk["ftail"]])
    assert (got != ref).sum() <= 1 and np.abs(got - ref).max() <= 1, (got, ref)


def test_duplicate_rows_accumulate(hip):
    """The same pair repeated 50 times: every copy's gradient must land (atomic scatter), loss is 50x."""
    rng = np.random.default_rng(3)
    hp = dict(hidden_size=24, l1_flag=False, margin=2.0)
    P, m, tr, cfg, _ = _setup(hip, "transe", hp, 20, 3, rng)
    one = (np.array([1]), np.array([2]), np.array([3]), np.array([1]), np.array([2]), np.array([7]))
    rep = tuple(np.repeat(a, 50) for a in one)
    l1 = tr.train_step_pairwise(*[hip.dev(a) for a in one]).item()
    g1 = tr.flat.grad.clone()
    tr.flat.grad.zero_()
    l50 = tr.train_step_pairwise(*[hip.dev(a) for a in rep]).item()
    assert np.isclose(l50, 50 * l1, rtol=1e-5)
    assert torch.allclose(tr.flat.grad, 50 * g1, rtol=1e-4, atol=1e-6)


def test_zero_rows_take_the_eps_branch_of_normalize(hip):
    rng = np.random.default_rng(4)
    hp = dict(hidden_size=16, l1_flag=True, margin=1.0)
    E, R = 12, 3
    P = ko.init_params("transe", rng, tot_entity=E, tot_relation=R, hidden_size=16)
    P["ent_embeddings"][5] = 0.0          # ||x|| < eps: x / eps, gradient g / eps
    P["rel_embeddings"][1] = 0.0
    m = hip.model_from_params("transe", P, hp, E, R)
    h, r, t = np.array([5, 1, 2, 5]), np.array([1, 1, 0, 2]), np.array([3, 5, 5, 5])
    with torch.no_grad():
        got = m(hip.dev(h), hip.dev(r), hip.dev(t)).cpu().numpy()
    assert np.allclose(got, ko.score("transe", P, h, r, t, l1_flag=True), atol=1e-5, rtol=1e-5)
    assert np.all(np.isfinite(got))


def test_bench_two_ranks_owner_computes_gradient():
    """The same with the data-parallel owner-computes gradient step forced on (KGE_PULL=1: kge_pull_step in gradient mode, no
    atomics, then reduce-scatter / sharded optimiser / all-gather / row norms): replicas identical, one JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KGE_BENCH_SHARE_GPU="1", KGE_BENCH_CHECK_REPLICAS="1", KGE_PULL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--batch", "4096", "--eval-triples", "256"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "owner-computes gradient" in d["config"]["step_path"]
    assert d["value"] > 0 and "REPLICAS_IDENTICAL 1" in out.stdout
    ph = d["phases_us"]     # per-phase event times of the data-parallel step (max over ranks)
    # (6.5 MB of tables: one all-reduce per step -- recorded under `reduce_scatter` -- and no parameter all-gather)
    assert all(ph[k] > 0 for k in ("compute", "reduce_scatter", "optimiser", "row_norms")) and ph["all_gather"] == 0, ph
    assert d["collectives"]["per_step"].startswith("all_reduce")


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N>1 path end to end on real kernels: two ranks (gloo, both on device 0) shard the batch and the Philox
    stream, all-reduce the flat gradient and must end with bit-identical tables; rank 0 prints the one JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # KGE_DP_ALLREDUCE=0: the sharded step larger tables take (reduce-scatter, optimiser on the rank's shard, all-gather)
    env = dict(os.environ, KGE_BENCH_SHARE_GPU="1", KGE_BENCH_CHECK_REPLICAS="1", KGE_DP_ALLREDUCE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--batch", "4096", "--eval-triples", "256"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8192
    assert d["value"] > 0 and d["eval"]["value"] > 0 and "cpu_baseline" not in d
    assert "REPLICAS_IDENTICAL 1" in out.stdout
    ph = d["phases_us"]
    assert all(ph[k] > 0 for k in ("compute", "reduce_scatter", "optimiser", "all_gather")), ph


@pytest.mark.parametrize("env_extra,label", [({}, "all-reduce"), ({"KGE_DP_ALLREDUCE": "0"}, "sharded"), ({"KGE_PULL": "1"}, "owner-computes")])
def test_bench_eight_ranks_share_one_gpu(env_extra, label):
    """The launch the driver makes on an 8-GPU node (`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`), rehearsed on ONE
    GPU: eight processes (gloo, all on device 0) shard the batch and the Philox stream eight ways, exchange the flat gradient, and must
    end with bit-identical tables; the test split (59 071 triples, not a multiple of 8) is sharded over the ranks for the eval leg; rank
    0 prints the one JSON line with the per-phase times and the collective description.  Nothing here needs RCCL: what is rehearsed
    is every line of bench.py's and the Trainer's N = 8 path except the transport."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KGE_BENCH_SHARE_GPU="1", KGE_BENCH_CHECK_REPLICAS="1", **env_extra)
    port = {"all-reduce": "29531", "sharded": "29533", "owner-computes": "29535"}[label]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2", "--batch", "4096"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8 * 4096 and d["config"]["parallelism"] == "dp8"
    assert d["value"] > 0 and d["repeats"] >= 7 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert "REPLICAS_IDENTICAL 1" in out.stdout and d["replicas_identical"] is True
    assert d["collectives"]["world_size"] == 8 and d["collectives"]["backend"] == "gloo"
    ph = d["phases_us"]
    assert set(ph) >= {"compute", "reduce_scatter", "optimiser", "all_gather", "row_norms"}, ph
    assert ph["compute"] > 0 and ph["reduce_scatter"] > 0 and ph["optimiser"] > 0
    assert (ph["all_gather"] > 0) == (label == "sharded"), ph      # 6.5 MB of tables: one all-reduce unless the sharded step is forced
    assert d["eval"]["test_triples"] == 59071 // 8 and d["eval"]["value"] > 0 and "cpu_baseline" not in d


@pytest.mark.parametrize("model,hp,opt,lr,fmr_drop,mrr_gain", [
    ("transe", dict(hidden_size=32, l1_flag=True, margin=1.0), "adam", 0.01, 0.5, 3.0),
    # a symmetric bilinear model can only partly fit a translation graph: it must still clearly beat chance
    ("distmult", dict(hidden_size=32, lmbda=1e-5), "adagrad", 0.1, 0.8, 1.5),
    # the per-relation matrices fit the small training set long before they generalise: a milder bar
    ("transr", dict(ent_hidden_size=32, rel_hidden_size=32, l1_flag=False, margin=1.0), "adam", 0.01, 0.75, 3.0)])
def test_train_model_learns_a_planted_graph(hip, model, hp, opt, lr, fmr_drop, mrr_gain):
    """Trainer.train_model end to end (epoch loop, hipGraph replay, mini_test + early stopper, full_test): on a graph
    generated by a planted translation model the filtered ranks must move far away from chance."""
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(7)
    E, R, dp = 400, 6, 8
    ent = rng.normal(size=(E, dp)); rel = rng.normal(size=(R, dp)) * 1.5
    trip = set()
    for h in range(E):
        for r in range(R):
            t = int(np.argmin(np.abs(ent[h] + rel[r] - ent).sum(1) + 1e9 * (np.arange(E) == h)))
            trip.add((h, r, t))
    trip = np.asarray(sorted(trip), dtype=np.int64)
    trip = trip[rng.permutation(len(trip))]
    n_test = 200
    train, valid, test = trip[2 * n_test:], trip[:n_test], trip[n_test:2 * n_test]
    cfg = hip.make_config(E, R, dict(hp, neg_rate=1), train, valid, test, optimizer=opt, lr=lr, batch_size=256)
    cfg.epochs, cfg.test_step, cfg.test_num, cfg.patience = 60, 20, 100, 10
    torch.manual_seed(0)
    m = hip.model_from_params(model, {}, hp, E, R, train=train)
    tr = Trainer(m, cfg)
    tr.build_model()
    before = tr.evaluator.test(test, n_test, epoch=0)
    tr.train_model()
    after = tr.evaluator.test(test, n_test, epoch=0)
    assert before["fmr"] > 0.3 * E                      # chance level at initialisation
    assert after["fmr"] < fmr_drop * before["fmr"], (before, after)
    assert after["fmrr"] > mrr_gain * before["fmrr"], (before, after)


def test_evaluator_without_cached_filter_dicts_groups_the_flat_splits(hip):
    """A KG cache carrying only the three splits: filters are built from them (vectorised CSR) -- same ranks."""
    from golden_util import Case
    from pykg2vec_amd.evaluator import Evaluator
    c = Case("transe_l1")
    m = hip.model_from_case(c, "adam.final.")
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    want = Evaluator(m, cfg).rank_all(c.test, len(c.test)).cpu().numpy()
    for k in ("hr_t", "tr_h"):
        del cfg.knowledge_graph.cache[k]
    got = Evaluator(m, cfg).rank_all(c.test, len(c.test)).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("model,hp,opt", [
    ("transe", dict(hidden_size=1500, l1_flag=True, margin=1.0), "adam"),
    ("transe", dict(hidden_size=2048, l1_flag=False, margin=1.0), "sgd"),
    ("distmult", dict(hidden_size=1100, lmbda=1e-4), "adagrad"),
    ("complex", dict(hidden_size=1028, lmbda=1e-4), "adagrad"),
    ("rotate", dict(hidden_size=1200, margin=6.0, alpha=1.0, neg_rate=4), "adam"),
    ("transh", dict(hidden_size=1040, l1_flag=True, margin=1.0), "sgd")])
def test_hidden_sizes_between_1024_and_2048(hip, model, hp, opt):
    """Round 5: rows of 1 025 .. 2 048 floats (the reference has no limit; rounds 1-4 refused them): forward energies, one fused training
    step (loss + updated tables through the dense optimiser) and filtered ranks against the numpy oracle."""
    from pykg2vec_amd.trainer import Trainer
    from pykg2vec_amd.evaluator import Evaluator
    rng = np.random.default_rng(3)
    E, R, B = 300, 9, 48
    neg = hp.get("neg_rate", 1)
    shape_kw = {k: v for k, v in hp.items() if k in ("hidden_size", "margin") and (k != "margin" or model == "rotate")}
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, **shape_kw)
    trip = np.stack([rng.integers(E, size=400), rng.integers(R, size=400), rng.integers(E, size=400)], 1)
    h, r, t = (trip[:64, i] for i in range(3))
    score_hp = {k: v for k, v in hp.items() if k not in ("neg_rate", "alpha")}
    m = hip.model_from_params(model, P, hp, E, R, train=trip)
    with torch.no_grad():
        got = m(hip.dev(h), hip.dev(r), hip.dev(t)).cpu().numpy()
    want = ko.score(model, P, h, r, t, **score_hp)
    assert np.allclose(got, want, atol=2e-4, rtol=2e-5), np.abs(got - want).max()
    # one training step on an explicit batch
    pos = trip[:B]
    nh, nt = np.repeat(pos[:, 0], neg).copy(), np.repeat(pos[:, 2], neg).copy()
    flip = rng.random(B * neg) > 0.5
    ent = rng.integers(E, size=B * neg)
    nh[~flip] = ent[~flip]; nt[flip] = ent[flip]
    nr = np.repeat(pos[:, 1], neg)
    pointwise = model in ("distmult", "complex")
    batch = ko.pointwise_layout(pos, nh, nr, nt, neg) if pointwise else (pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt)
    loss_ref, G, _, _ = ko.train_step_grads(model, P, batch, **hp)
    cfg = hip.make_config(E, R, hp, trip[:300], trip[300:350], trip[350:], optimizer=opt, lr=0.01, batch_size=B)
    tr = Trainer(m, cfg)
    tr.build_model()
    dev_batch = [hip.dev(x) for x in batch]
    loss = (tr.train_step_pointwise if pointwise else tr.train_step_pairwise)(*dev_batch)
    assert np.isclose(loss.item(), loss_ref, rtol=2e-4), (loss.item(), loss_ref)
    for name, gview in zip(P, tr.flat.grad_views):
        g = gview.cpu().numpy()
        scale = max(1e-3, float(np.abs(G[name]).max()))
        assert np.allclose(g, G[name], atol=2e-4 * scale, rtol=2e-3), (name, np.abs(g - G[name]).max(), scale)
    tr._reduce_and_step()
    # filtered ranks of a few test triples
    Pn = {k: p.detach().cpu().numpy() for k, p in zip(P, [v for v in tr.flat.views])}
    q = trip[350:358]
    ranks = Evaluator(m, cfg).rank_all(q, len(q)).cpu().numpy()
    hr_t, tr_h = cfg.knowledge_graph.cache["hr_t"], cfg.knowledge_graph.cache["tr_h"]
    _, ref = ko.evaluate(model, Pn, q, hr_t, tr_h, **score_hp)
    ref = np.stack([ref["head"], ref["tail"], ref["fhead"], ref["ftail"]])
    assert np.abs(ranks - ref).max() <= 2 and (ranks != ref).mean() <= 0.2, (ranks, ref)
