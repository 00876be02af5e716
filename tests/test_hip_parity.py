"""Parity of the HIP path (through the C ABI) with (a) the frozen outputs of the live reference in tests/golden
and (b) the CPU oracle on seeded inputs of other shapes.  fp32 tolerance: atol 1e-5 + rtol 1e-5 on energies
(BASELINE.json north_star); integer ranks exact given equal scores."""
import numpy as np
import pytest
import torch

import kge_oracle as ko
from golden_util import CASES, Case, close, rank_band_ok

pytestmark = pytest.mark.gpu

VECTOR_CASES = list(CASES)
EVAL_CASES = list(CASES)
GRAD_TOL = dict(atol=2e-5, rtol=1e-4)


def assert_ranks_inside_band(case, ranks, ref, scores, trips):
    """Every rank that differs from the reference's must be explained by candidates inside the fp32 tolerance band
    around the true candidate's energy (golden_util.rank_band_ok); the observed agreement is appended to
    gpurun_out/rank_agreement.json (kept under profiles/ per round)."""
    import json
    import os
    rep = {"queries": 2 * len(trips), "raw_equal": 0, "filtered_equal": 0, "max_abs_rank_diff": 0, "flips": []}
    for i, (h, r, t) in enumerate(trips):
        for side, row, true, a, b in (("tail", scores[2 * i], int(t), 1, 3), ("head", scores[2 * i + 1], int(h), 0, 2)):
            ok_r, near = rank_band_ok(row, true, ranks[a, i], ref[a, i])
            ok_f, _ = rank_band_ok(row, true, ranks[b, i], ref[b, i])
            assert ok_r and ok_f, (case, i, side, ranks[:, i], ref[:, i], near)
            rep["raw_equal"] += int(ranks[a, i] == ref[a, i])
            rep["filtered_equal"] += int(ranks[b, i] == ref[b, i])
            rep["max_abs_rank_diff"] = max(rep["max_abs_rank_diff"], abs(int(ranks[a, i]) - int(ref[a, i])))
            if ranks[a, i] != ref[a, i] or ranks[b, i] != ref[b, i]:
                rep["flips"].append({"triple": i, "side": side, "gpu": [int(ranks[a, i]), int(ranks[b, i])],
                                     "reference": [int(ref[a, i]), int(ref[b, i])], "candidates_inside_band": near})
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "rank_agreement.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[case] = rep
    json.dump(doc, open(path, "w"), indent=1)
    return rep


@pytest.fixture(scope="module")
def hip():
    import hip_util
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return hip_util


@pytest.mark.parametrize("name", VECTOR_CASES)
def test_forward_matches_reference_golden(hip, name):
    c = Case(name)
    m = hip.model_from_case(c)
    b = c.batch(0)
    with torch.no_grad():
        if c.pointwise:
            got = m(hip.dev(b[0]), hip.dev(b[1]), hip.dev(b[2])).cpu().numpy()
            assert close(got, c.z["scores0"]), np.abs(got - c.z["scores0"]).max()
        else:
            gp = m(hip.dev(b[0]), hip.dev(b[1]), hip.dev(b[2])).cpu().numpy()
            gn = m(hip.dev(b[3]), hip.dev(b[4]), hip.dev(b[5])).cpu().numpy()
            assert close(gp, c.z["scores0_pos"]) and close(gn, c.z["scores0_neg"])


@pytest.mark.parametrize("name", VECTOR_CASES)
def test_autograd_path_matches_reference_grads(hip, name):
    """model.forward + Criterion loss + loss.backward(): the path the UNMODIFIED reference Trainer drives
    (utils/trainer.py:147-180,298)."""
    c = Case(name)
    m = hip.model_from_case(c)
    b = [hip.dev(x) for x in c.batch(0)]
    m.train()
    if c.pointwise:
        preds = m(b[0], b[1], b[2])
        loss = m.loss(preds, b[3].type(preds.type())) + m.get_reg(b[0], b[1], b[2])
    else:
        pos, neg = m(b[0], b[1], b[2]), m(b[3], b[4], b[5])
        if m.model_name == "rotate":
            loss = m.loss(pos, neg, c.hp["neg_rate"], c.hp["alpha"])
        else:
            loss = m.loss(pos, neg, c.hp["margin"])
        loss = loss + m.get_reg(None, None, None)
    loss.backward()
    assert close(loss.item(), c.z["loss0"]), (loss.item(), c.z["loss0"])
    for k, p in hip.table_parameters(m):
        if "grad0." + k not in c.z.files:  # a table forward never reads (QuatE rel_w): the reference leaves grad None
            assert p.grad is None or not p.grad.any()
            continue
        ref = c.z["grad0." + k]
        got = p.grad.cpu().numpy()
        assert np.allclose(got, ref, **GRAD_TOL), (k, np.abs(got - ref).max())


@pytest.mark.parametrize("name", VECTOR_CASES)
def test_fused_step_matches_reference_loss_and_grads(hip, name):
    from pykg2vec_amd.trainer import Trainer
    c = Case(name)
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    m = hip.model_from_case(c)
    tr = Trainer(m, cfg)
    tr.build_model()
    b = [hip.dev(x) for x in c.batch(0)]
    loss = tr.train_step_pointwise(*b) if c.pointwise else tr.train_step_pairwise(*b)
    assert close(loss.item(), c.z["loss0"], atol=2e-5, rtol=2e-5), (loss.item(), c.z["loss0"])
    for p, g in zip(m.parameter_list, tr.flat.grad_views):
        name_ = [n for n, q in m.named_parameters() if q is p.weight][0]
        if "grad0." + name_ not in c.z.files:
            assert not g.any()
            continue
        ref = c.z["grad0." + name_]
        assert np.allclose(g.cpu().numpy(), ref, **GRAD_TOL), (name_, np.abs(g.cpu().numpy() - ref).max())


@pytest.mark.parametrize("opt", ["sgd", "adam", "adagrad", "rms"])
@pytest.mark.parametrize("name", VECTOR_CASES)
def test_three_fused_training_steps_match_reference_weights(hip, name, opt):
    from pykg2vec_amd.trainer import Trainer
    c = Case(name)
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test, optimizer=opt, lr=0.05)
    m = hip.model_from_case(c)
    tr = Trainer(m, cfg)
    tr.build_model()
    losses = []
    for s in range(3):
        b = [hip.dev(x) for x in c.batch(s)]
        loss = tr.train_step_pointwise(*b) if c.pointwise else tr.train_step_pairwise(*b)
        tr._reduce_and_step()
        losses.append(loss.item())
    assert close(np.asarray(losses), c.z["%s.losses" % opt], atol=3e-5, rtol=3e-5)
    tol = 2e-3 if opt == "rms" else 1e-4  # see tests/test_oracle_golden.py on RMSprop's noise amplification
    for k, p in hip.table_parameters(m):
        ref = c.z["%s.final.%s" % (opt, k)]
        got = p.detach().cpu().numpy()
        if opt == "rms":
            # RMSprop's first steps move a weight by ~10*lr*sign(g) however small g is, so an entry whose gradient is a
            # pure rounding residue of cancelling contributions (order-dependent under float atomics) may land
            # elsewhere; such entries are isolated -- everything else must match
            bad = np.abs(got - ref) > tol + 1e-4 * np.abs(ref)
            assert bad.mean() < 2e-3, (k, bad.sum(), np.abs(got - ref).max())
            continue
        assert np.allclose(got, ref, atol=tol, rtol=1e-4), (k, np.abs(got - ref).max())


def test_tie_policy_on_clamp_saturated_simple(hip):
    """Exact ties (SimplE's +-20 clamp, tests/golden/ref_simple_ties.npz from the live reference): the HIP ranks are the
    count of STRICTLY lower energies -- the optimistic end of the true candidate's tie group -- and the reference's topk scan
    (utils/evaluator.py:70-123) lands inside the group: less <= reference <= less + ties.  (behavioural
    differences (c).)"""
    from golden_util import tie_bracket
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    c = Case("simple_ties")
    m = hip.model_from_case(c)
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    n = len(c.z["eval.rank_head"])
    ranks, tie_counts = Evaluator(m, cfg).rank_all(c.test, n, return_ties=True)
    ranks, tie_counts = ranks.cpu().numpy(), tie_counts.cpu().numpy()        # tie_counts [2, n]: head sweeps, tail sweeps
    scores = K.eval_sweep_scores(m.make_desc(), hip.dev(c.test[:n])).cpu().numpy()
    ref_sw = c.z["eval.sweeps"]
    sat = np.abs(ref_sw) == 20.0
    assert np.array_equal(scores[sat], ref_sw[sat]) and close(scores, ref_sw, atol=3e-4)
    hr_t, tr_h = c.filters()
    ref = np.stack([c.z["eval.rank_head"], c.z["eval.rank_tail"], c.z["eval.frank_head"], c.z["eval.frank_tail"]])
    ties_total = 0
    for i, (h, r, t) in enumerate(c.test[:n]):
        h, r, t = int(h), int(r), int(t)
        for row, true, known, a, b in ((scores[2 * i], t, hr_t[(h, r)], 1, 3), (scores[2 * i + 1], h, tr_h[(t, r)], 0, 2)):
            less, ties, fless, fties = tie_bracket(row, true, known)
            assert (ranks[a, i], ranks[b, i]) == (less, fless)            # count-based, exact function of the GPU scores
            assert tie_counts[a, i] == ties                                # ... and the sweep reports the size of the tie group
            near = int(np.sum((np.abs(row - row[true]) <= 6e-4) & (row != row[true])))
            assert less - near <= ref[a, i] <= less + ties + near and fless - near <= ref[b, i] <= fless + fties + near, \
                (i, less, ties, fless, fties, ref[:, i], near)
            ties_total += ties
    assert ties_total > 100
    # the same counts from the matrix-core sweep and from the VALU sweep (both count while they sweep); Evaluator.test() warns
    for gemm in (0, 1):
        K.set_switch("EVAL_GEMM", gemm)
        try:
            r2, t2 = Evaluator(m, cfg).rank_all(c.test, n, return_ties=True)
        finally:
            K.set_switch("EVAL_GEMM", None)
        assert np.array_equal(r2.cpu().numpy(), ranks) and np.array_equal(t2.cpu().numpy(), tie_counts), gemm
    ev = Evaluator(m, cfg)
    ev.test(c.test, n, epoch=0)
    assert ev.tie_counts is not None and int(ev.tie_counts.sum()) == int(tie_counts.sum())


def test_no_ties_reported_for_an_unsaturated_model(hip):
    """A distance model with generic weights has no exact ties: the tie report is all zeros (and costs nothing to ask for)."""
    from pykg2vec_amd.evaluator import Evaluator
    c = Case("transe_l1")
    m = hip.model_from_case(c, "adam.final.")
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    ranks, ties = Evaluator(m, cfg).rank_all(c.test, 8, return_ties=True)
    assert int(ties.abs().sum()) == 0 and torch.equal(ranks, Evaluator(m, cfg).rank_all(c.test, 8))


@pytest.mark.parametrize("name", EVAL_CASES)
def test_one_sided_sweeps_and_rank_hooks(hip, name):
    """kge_eval_sweep_scores_side (what the predict_tail_rank / predict_head_rank hooks of utils/evaluator.py:250-252,263-265 call):
    bit-identical to the corresponding rows of the two-sided sweep, whatever sits in the column the side does not read; the hooks
    return the ids in ascending-energy order of that row."""
    from pykg2vec_amd import kernels as K
    c = Case(name)
    m = hip.model_from_case(c, "adam.final.")
    if c.model == "rescal":
        m.normalize_tables()
    trips = c.test[:5] if c.model != "transr" else c.test[:1]
    both = K.eval_sweep_scores(m.make_desc(), hip.dev(trips))
    junk = trips.copy()
    junk[:, 2] = (junk[:, 2] + 7) % c.E
    tail = K.eval_sweep_scores_side(m.make_desc(), hip.dev(junk if c.model not in ("transr", "ntn") else trips), 0)
    junk = trips.copy()
    junk[:, 0] = (junk[:, 0] + 3) % c.E
    head = K.eval_sweep_scores_side(m.make_desc(), hip.dev(junk if c.model not in ("transr", "ntn") else trips), 1)
    assert torch.equal(tail, both[0::2]) and torch.equal(head, both[1::2])
    h, r, t = (hip.dev(trips[:1, i]) for i in range(3))
    ids = m.predict_tail_rank(h, r, topk=c.E)
    assert ids.shape == (1, c.E)
    # torch.topk = largest first, exactly what the reference's hook convention returns (models/projection.py:119-125)
    assert torch.equal(both[0][ids[0]], torch.sort(both[0], descending=True).values)
    ids = m.predict_head_rank(t, r, topk=c.E)
    assert torch.equal(both[1][ids[0]], torch.sort(both[1], descending=True).values)


def test_one_sided_sweep_on_the_matrix_core_path(hip):
    """>= 512 query rows: the dot-product forms sweep on k_eval_gemm; one-sided, the 600 wanted rows alone fill the tiles."""
    from pykg2vec_amd import kernels as K
    c = Case("complex")
    m = hip.model_from_case(c, "adam.final.")
    rng = np.random.default_rng(0)
    trips = np.stack([rng.integers(c.E, size=600), rng.integers(c.R, size=600), rng.integers(c.E, size=600)], 1)
    both = K.eval_sweep_scores(m.make_desc(), hip.dev(trips))
    for side in (0, 1):
        assert torch.equal(K.eval_sweep_scores_side(m.make_desc(), hip.dev(trips), side), both[side::2])


@pytest.mark.parametrize("name", EVAL_CASES)
def test_eval_sweep_scores_and_ranks_match_reference(hip, name):
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    c = Case(name)
    m = hip.model_from_case(c, "adam.final.")
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    if c.model == "rescal":
        m.normalize_tables()  # the reference's forward renormalises the tables before every sweep (pairwise.py:843-844)
    def sweep_scores(trips):
        if c.model == "transr":  # one relation per sweep call (the candidate table is projected by M_r)
            return np.concatenate([K.eval_sweep_scores(m.make_desc(), hip.dev(trips[i:i + 1])).cpu().numpy()
                                   for i in range(len(trips))])
        return K.eval_sweep_scores(m.make_desc(), hip.dev(trips)).cpu().numpy()

    sw = sweep_scores(c.test[:4])
    assert close(sw, c.z["eval.sweeps"], atol=2e-5, rtol=2e-5), np.abs(sw - c.z["eval.sweeps"]).max()
    ev = Evaluator(m, cfg)
    n = len(c.z["eval.rank_head"])
    ranks = ev.rank_all(c.test, n).cpu().numpy()
    ref = np.stack([c.z["eval.rank_head"], c.z["eval.rank_tail"], c.z["eval.frank_head"], c.z["eval.frank_tail"]])
    # ranks are exact functions of OUR fp32 scores (checked below); against the reference they may differ only
    # where two candidates are closer than the fp32 tolerance band
    scores = sweep_scores(c.test[:n])
    hr_t, tr_h = c.filters()
    for i, (h, r, t) in enumerate(c.test[:n]):
        rt = ko.rank_from_scores(scores[2 * i], int(t), hr_t[(int(h), int(r))])
        rh = ko.rank_from_scores(scores[2 * i + 1], int(h), tr_h[(int(t), int(r))])
        assert (ranks[1, i], ranks[3, i]) == rt and (ranks[0, i], ranks[2, i]) == rh
    rep = assert_ranks_inside_band(name, ranks, ref, scores, c.test[:n])
    assert rep["raw_equal"] + rep["filtered_equal"] >= 4 * n - 2, rep   # and near-ties are rare
    metrics = ev.test(c.test, n, epoch=0)
    assert np.isclose(metrics["fmr"], c.z["eval.fmr"], rtol=0.02)
    if c.model in ("transh", "transd"):
        # grouped evaluation (candidates transformed once per relation, then the plain sweep) against the in-sweep transform
        # and the reference: different rounding, so only fp32 near-ties may move
        ev3 = Evaluator(m, cfg)
        ev3.GROUPED_MIN_TRIPLES_PER_RELATION = 0
        r3 = ev3.rank_all(c.test, n).cpu().numpy()
        assert ev3._fingerprint(c.test, n) in ev3._groups
        assert np.abs(r3 - ranks).max() <= 1 and (r3 != ranks).sum() <= 2, (r3, ranks)
        assert np.abs(r3 - ref).max() <= 1 and (r3 != ref).sum() <= 2, (r3, ref)
        ev3.TABLE_BUDGET_BYTES = 1  # one relation group per call
        ev4 = Evaluator(m, cfg)
        ev4.GROUPED_MIN_TRIPLES_PER_RELATION, ev4.TABLE_BUDGET_BYTES = 0, 1
        assert np.array_equal(ev4.rank_all(c.test, n).cpu().numpy(), r3)
        ev5 = Evaluator(m, cfg)  # mixed: relations with >= 2 test triples grouped, the rest through the in-sweep transform
        ev5.GROUPED_MIN_TRIPLES_PER_RELATION = 2
        r5 = ev5.rank_all(c.test, n).cpu().numpy()
        nd5 = ev5._groups[ev5._fingerprint(c.test, n)][2]
        assert 0 < nd5 < n
        assert np.abs(r5 - ranks).max() <= 1 and (r5 != ranks).sum() <= 2
    if c.model == "transr":  # relation groups split over several grouped calls (tiny table budget) give the same ranks
        ev2 = Evaluator(m, cfg)
        ev2.TABLE_BUDGET_BYTES = 1
        assert np.array_equal(ev2.rank_all(c.test, n).cpu().numpy(), ranks)
        assert len(ev2._groups[ev2._fingerprint(c.test, n)][1]) > 1


def test_pretrained_fb15k_transe_slice(hip):
    import os
    from golden_util import GOLDEN
    from pykg2vec_amd.evaluator import Evaluator
    z = np.load(os.path.join(GOLDEN, "ref_pretrained_transe_fb15k.npz"))
    P = {"ent_embeddings": z["init.ent_embeddings.weight"], "rel_embeddings": z["init.rel_embeddings.weight"]}
    for l1, key in ((True, "l1"), (False, "l2")):
        hp = dict(hidden_size=50, l1_flag=l1, margin=1.0)
        m = hip.model_from_params("transe", P, hp, int(z["E"]), int(z["R"]))
        with torch.no_grad():
            got = m(hip.dev(z["ids.h"]), hip.dev(z["ids.r"]), hip.dev(z["ids.t"])).cpu().numpy()
        assert close(got, z["scores_" + key]), np.abs(got - z["scores_" + key]).max()
        cfg = hip.make_config(int(z["E"]), int(z["R"]), hp, z["train"], z["valid"], z["test"])
        n = len(z["eval_%s.rank_head" % key])
        ranks = Evaluator(m, cfg).rank_all(z["test"], n).cpu().numpy()
        ref = np.stack([z["eval_%s.%s" % (key, k)] for k in ("rank_head", "rank_tail", "frank_head", "frank_tail")])
        from pykg2vec_amd import kernels as K
        scores = K.eval_sweep_scores(m.make_desc(), hip.dev(z["test"][:n])).cpu().numpy()
        rep = assert_ranks_inside_band("pretrained_fb15k_transe_" + key, ranks, ref, scores, z["test"][:n])
        assert rep["raw_equal"] + rep["filtered_equal"] >= 4 * n - 3, rep


SHAPES = [("transe", dict(hidden_size=50, l1_flag=True), 1), ("transe", dict(hidden_size=100, l1_flag=False), 1),
          ("transe", dict(hidden_size=200, l1_flag=True), 1), ("transe", dict(hidden_size=300, l1_flag=True), 1),
          ("transh", dict(hidden_size=100, l1_flag=True), 1), ("transd", dict(ent_hidden_size=64, rel_hidden_size=64, l1_flag=False), 1),
          ("rotate", dict(hidden_size=1000, margin=24.0, neg_rate=16, alpha=1.0), 16),
          ("rotate", dict(hidden_size=33, margin=6.0, neg_rate=3, alpha=0.5), 3),
          ("rotate", dict(hidden_size=20, margin=6.0, neg_rate=40, alpha=1.0), 40),   # > lane group: three-launch path
          ("distmult", dict(hidden_size=100, lmbda=0.01), 2), ("complex", dict(hidden_size=200, lmbda=1e-4), 1),
          ("complexn3", dict(hidden_size=37, lmbda=0.05), 2), ("analogy", dict(hidden_size=200, lmbda=0.01), 1),
          ("rescal", dict(hidden_size=50), 1), ("rescal", dict(hidden_size=200), 1), ("rescal", dict(hidden_size=33), 1),
          ("ntn", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4), 1),
          ("ntn", dict(ent_hidden_size=40, rel_hidden_size=33, lmbda=0.1), 1),
          ("transm", dict(hidden_size=50, l1_flag=False), 1), ("transm", dict(hidden_size=300, l1_flag=True), 1),
          ("transr", dict(ent_hidden_size=50, rel_hidden_size=50, l1_flag=True), 1),
          ("transr", dict(ent_hidden_size=128, rel_hidden_size=100, l1_flag=False), 1),
          ("transr", dict(ent_hidden_size=33, rel_hidden_size=70, l1_flag=True), 1),
          ("cp", dict(hidden_size=50, lmbda=1e-4), 1), ("cp", dict(hidden_size=600, lmbda=1e-4), 2),
          ("simple", dict(hidden_size=100, lmbda=0.1), 1), ("simple_ignr", dict(hidden_size=100, lmbda=0.1), 3),
          ("quate", dict(hidden_size=100, lmbda=0.2), 1), ("quate", dict(hidden_size=200, lmbda=0.1), 2),
          ("quate", dict(hidden_size=300, lmbda=0.05), 1),
          # widest register geometry (64 lanes x 16 chunks) for the second model group
          ("quate", dict(hidden_size=520, lmbda=0.05), 1), ("simple", dict(hidden_size=700, lmbda=0.1), 1),
          ("transm", dict(hidden_size=1000, l1_flag=False), 1)]


@pytest.mark.parametrize("model,hp,neg_rate", SHAPES)
def test_scores_loss_grads_vs_oracle_other_shapes(hip, model, hp, neg_rate):
    """Every (G, NCH) register geometry of the row kernels, against the oracle on seeded inputs."""
    _check_step_vs_oracle(hip, model, hp, neg_rate, 300, 11, 160)


@pytest.mark.parametrize("d,kr,B", [(100, 100, 300), (40, 33, 70), (50, 7, 129), (128, 128, 64), (17, 1, 200), (96, 64, 1100)])
def test_ntn_gemm_forms_vs_oracle(hip, d, kr, B, monkeypatch):
    """The large-batch NTN kernels (k_ntn_rows / k_ntn_outer: batch-as-M GEMMs on v_mfma_f32_16x16x4_f32, default from
    1 024 rows on), forced on for oracle-sized batches: every block-count instantiation class (1 ... 8 blocks of 16, widths
    that are not multiples of 4 or 16), partial 128-triple tiles, several slice groups and split-K chunks."""
    monkeypatch.setenv("KGE_NTN_BIG", "1")
    # (the bias gradient is a sum of 2 B terms of either sign: its fp32 error against the float64 oracle grows with the batch,
    # in the small-batch kernels as well -- 5.7e-5 there, 8.8e-5 here at B = 1 100)
    _check_step_vs_oracle(hip, "ntn", dict(ent_hidden_size=d, rel_hidden_size=kr, lmbda=1e-3), 1, 300, 11, B,
                          grad_atol=5e-5 if B < 1000 else 2e-4)


def _check_step_vs_oracle(hip, model, hp, neg_rate, E, R, B, grad_atol=5e-5):
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(42)
    shape_kw = {k: v for k, v in hp.items() if k in ("hidden_size", "ent_hidden_size", "rel_hidden_size", "margin")}
    if model != "rotate":
        shape_kw.pop("margin", None)
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, **shape_kw)
    hp = dict(hp)
    hp.setdefault("margin", 1.0)
    pos = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
    nh = np.repeat(pos[:, 0], neg_rate); nr = np.repeat(pos[:, 1], neg_rate); nt = np.repeat(pos[:, 2], neg_rate)
    flip = rng.random(B * neg_rate) > 0.5
    rnd = rng.integers(E, size=B * neg_rate)
    nh = np.where(flip, nh, rnd); nt = np.where(flip, rnd, nt)
    pointwise = model in ko.POINTWISE
    if pointwise:
        batch = ko.pointwise_layout(pos, nh, nr, nt, neg_rate)
    else:
        batch = (pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt)
    hp_run = dict(hp, neg_rate=neg_rate)
    if model == "transm":
        hp_run["theta"] = ko.transm_theta(pos, R)
    if model in ("simple", "simple_ignr"):  # inflate some rows so the +-20 clamp (and its zero gradient) is exercised
        for k in P:
            P[k][:40] *= 12.0
    loss_ref, G_ref, scores_ref, _ = ko.train_step_grads(model, P, batch, **hp_run)
    m = hip.model_from_params(model, P, hp, E, R, train=pos)
    cfg = hip.make_config(E, R, hp_run, pos, pos[:1], pos[:1])
    tr = Trainer(m, cfg)
    tr.build_model()
    b = [hip.dev(x) for x in batch]
    with torch.no_grad():
        s0 = m(b[0], b[1], b[2]).cpu().numpy()
    assert close(s0, scores_ref[0], atol=2e-5, rtol=2e-5), np.abs(s0 - scores_ref[0]).max()
    loss = tr.train_step_pointwise(*b) if pointwise else tr.train_step_pairwise(*b)
    assert np.isclose(loss.item(), loss_ref, rtol=5e-5, atol=5e-5), (loss.item(), loss_ref)
    names = [n.split(".")[0] for n, _ in hip.table_parameters(m)]
    for nme, g in zip(names, tr.flat.grad_views):
        got = g.cpu().numpy()
        scale = max(1.0, np.abs(G_ref[nme]).max())
        assert np.allclose(got, G_ref[nme], atol=grad_atol * scale, rtol=2e-4), (nme, np.abs(got - G_ref[nme]).max())


@pytest.mark.parametrize("k,E,R,B,margin", [(200, 300, 11, 160, 1.0), (64, 300, 3, 333, 1.0), (256, 50, 1, 40, 2.0), (4, 300, 40, 160, 1.0),
                                              (200, 3000, 400, 1024, 0.5), (120, 300, 11, 160, 0.02), (36, 9, 2, 1, 1.0),
                                              (50, 300, 11, 160, 1.0), (6, 40, 3, 70, 1.0), (250, 30, 2, 33, 1.0),      # k % 4 != 0: rows as float2
                                              (50, 3000, 400, 9000, 1.0),
                                              # >= 8192 pairs: dL/denergy left behind, relation-matrix gradient by the relation-owner
                                              # launch (k_rescal_pair_gm); > 16384: the block-aggregated grouping kernels
                                              (64, 3000, 5, 9000, 1.0), (32, 5000, 7, 20000, 1.0), (32, 5000, 700, 20000, 1.0),
                                              # many relations: the split form (k_rescal_rows + k_rescal_g2) from 512 pairs on
                                              (64, 3000, 600, 2000, 1.0), (50, 2000, 900, 700, 1.0)])
def test_rescal_pair_step_in_one_launch_matches_oracle(hip, monkeypatch, k, E, R, B, margin):
    """nr IS pr (one buffer): kge_train_pairwise_hinge groups PAIRS by relation and runs scores, hinge and the three gradients in
    one launch (k_rescal_pair).  Against the oracle, and against the three-launch path on the same batch (margin 0.02: most
    pairs inside the margin, whole tiles without gradient)."""
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(k + B)
    P = ko.init_params("rescal", rng, tot_entity=E, tot_relation=R, hidden_size=k)
    hp = dict(hidden_size=k, margin=margin, neg_rate=1)
    pos = np.stack([rng.integers(E, size=B), rng.integers(R, size=B) if B > 200 else np.sort(rng.integers(R, size=B)),
                    rng.integers(E, size=B)], 1)
    flip = rng.random(B) > 0.5
    rnd = rng.integers(E, size=B)
    nh = np.where(flip, pos[:, 0], rnd); nt = np.where(flip, rnd, pos[:, 2])
    batch = (pos[:, 0], pos[:, 1], pos[:, 2], nh, pos[:, 1], nt)
    loss_ref, G_ref, _, _ = ko.train_step_grads("rescal", P, batch, **hp)
    out = {}
    monkeypatch.delenv("KGE_RESCAL_UNFUSED", raising=False)
    for fused in (True, False):
        m = hip.model_from_params("rescal", P, hp, E, R, train=pos)
        cfg = hip.make_config(E, R, hp, pos, pos[:1], pos[:1])
        tr = Trainer(m, cfg)
        tr.build_model()
        b = [hip.dev(x) for x in batch]
        # (a separate nr buffer with the same contents: the three-launch path)
        loss = tr.train_step_pairwise(b[0], b[1], b[2], b[3], b[1] if fused else b[4], b[5])
        assert np.isclose(loss.item(), loss_ref, rtol=5e-5, atol=5e-5), (fused, loss.item(), loss_ref)
        out[fused] = [g.cpu().numpy().copy() for g in tr.flat.grad_views]
        for nme, got in zip(["ent_embeddings", "rel_matrices"], out[fused]):
            scale = max(1.0, np.abs(G_ref[nme]).max())
            assert np.allclose(got, G_ref[nme], atol=5e-5 * scale, rtol=2e-4), (fused, nme, np.abs(got - G_ref[nme]).max())
    for a, c in zip(out[True], out[False]):
        assert np.allclose(a, c, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("de,dr,E,R,B,l1,margin", [(100, 100, 300, 11, 160, True, 1.0), (50, 50, 300, 3, 333, False, 1.0),
                                                     (128, 100, 50, 1, 40, False, 2.0), (33, 70, 300, 40, 160, True, 1.0),
                                                     (100, 100, 3000, 400, 1500, True, 0.5), (64, 64, 300, 11, 160, True, 0.02),
                                                     (20, 7, 9, 2, 1, True, 1.0), (16, 16, 300, 11, 700, False, 1.0),
                                                     (70, 33, 40, 3, 70, False, 1.0), (128, 128, 30, 2, 33, True, 1.0),
                                                     (48, 80, 2000, 5, 3000, True, 1.0), (100, 100, 5000, 700, 9000, False, 1.0)])
def test_transr_pair_step_rows_matches_oracle(hip, monkeypatch, de, dr, E, R, B, l1, margin):
    """nr IS pr (one buffer): the large-batch TransR step (csrc/kge_transr_rows.hip: k_transr_rows + k_transr_g, batch-as-M GEMMs on
    v_mfma_f32_16x16x4_f32), forced on at oracle sizes -- every block-count class, d_e != d_r both ways, partial tiles and waves,
    relations spanning several runs (atomic G) and single-run ones (plain read-modify-write), margin 0.02 (whole workgroups without
    gradient).  Against the oracle and against the tile kernels (KGE_TRANSR_ROWS=0) on the same batch."""
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(de + 7 * dr + B)
    P = ko.init_params("transr", rng, tot_entity=E, tot_relation=R, ent_hidden_size=de, rel_hidden_size=dr)
    hp = dict(ent_hidden_size=de, rel_hidden_size=dr, l1_flag=l1, margin=margin, neg_rate=1)
    pos = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
    flip = rng.random(B) > 0.5
    rnd = rng.integers(E, size=B)
    nh = np.where(flip, pos[:, 0], rnd); nt = np.where(flip, rnd, pos[:, 2])
    batch = (pos[:, 0], pos[:, 1], pos[:, 2], nh, pos[:, 1], nt)
    loss_ref, G_ref, _, _ = ko.train_step_grads("transr", P, batch, **hp)
    out = {}
    for rows in (1, 2, 0):      # 1: M_r resident in LDS (the default form), 2: slab staging, 0: the tile kernels
        monkeypatch.setenv("KGE_TRANSR_ROWS", str(rows))
        m = hip.model_from_params("transr", P, hp, E, R, train=pos)
        cfg = hip.make_config(E, R, hp, pos, pos[:1], pos[:1])
        tr = Trainer(m, cfg)
        tr.build_model()
        b = [hip.dev(x) for x in batch]
        loss = tr.train_step_pairwise(b[0], b[1], b[2], b[3], b[1], b[5])
        assert np.isclose(loss.item(), loss_ref, rtol=5e-5, atol=5e-5), (rows, loss.item(), loss_ref)
        out[rows] = [g.cpu().numpy().copy() for g in tr.flat.grad_views]
        names = [n.split(".")[0] for n, _ in hip.table_parameters(m)]
        for nme, got in zip(names, out[rows]):
            scale = max(1.0, np.abs(G_ref[nme]).max())
            assert np.allclose(got, G_ref[nme], atol=5e-5 * scale, rtol=2e-4), (rows, nme, np.abs(got - G_ref[nme]).max())
    for k in (1, 2):
        for a, c in zip(out[k], out[0]):
            assert np.allclose(a, c, atol=5e-5, rtol=2e-4)


def test_missing_gpu_tensor_fails_loudly(hip):
    import pykg2vec_amd.pairwise as pw
    from pykg2vec_amd._lib import KgeHipError
    m = pw.TransE(tot_entity=10, tot_relation=3, hidden_size=8, l1_flag=True)  # left on the CPU
    with pytest.raises(KgeHipError):
        m(torch.tensor([1]), torch.tensor([1]), torch.tensor([1]))


def test_sampler_invariants_and_determinism(hip):
    from pykg2vec_amd import kernels as K
    c = Case("transe_l1")
    train = hip.dev(c.train)
    slots = K.triple_set_build(train)
    train_set = {tuple(map(int, x)) for x in c.train}
    ph, pr, pt = train[:, 0].contiguous(), train[:, 1].contiguous(), train[:, 2].contiguous()
    nh, nr, nt = K.corrupt(ph, pr, pt, 4, c.E, None, slots, seed=7, offset=0)
    nh2, _, nt2 = K.corrupt(ph, pr, pt, 4, c.E, None, slots, seed=7, offset=0)
    assert torch.equal(nh, nh2) and torch.equal(nt, nt2)
    NH, NR, NT = nh.cpu().numpy(), nr.cpu().numpy(), nt.cpu().numpy()
    tails = 0
    for j in range(len(NH)):
        h, r, t = c.train[j // 4]
        assert NR[j] == r and (NH[j] == h or NT[j] == t)
        assert (int(NH[j]), int(NR[j]), int(NT[j])) not in train_set
        assert 0 <= NH[j] < c.E and 0 <= NT[j] < c.E
        tails += int(NH[j] == h and NT[j] != t)
    frac = tails / len(NH)
    assert 0.42 < frac < 0.58, frac  # uniform sampling: P(corrupt tail) = 0.5
    bern = torch.full((c.R,), 0.9, device="cuda")  # prob 0.9 -> tail corrupted when u > 0.9
    nh3, _, nt3 = K.corrupt(ph, pr, pt, 4, c.E, bern, slots, seed=7, offset=0)
    frac_t = float((nh3.view(-1, 4) == ph.view(-1, 1)).float().mean())
    assert 0.05 < frac_t < 0.2, frac_t


@pytest.mark.parametrize("name,opt", [("transe_l1", "adam"), ("distmult", "adagrad"), ("rotate", "adam"), ("rescal", "sgd"),
                                      ("transm_l2", "sgd"), ("simple", "adagrad"), ("quate", "adagrad"), ("transr_l1", "sgd")])
def test_graph_replayed_epochs_equal_eager_epochs(hip, name, opt):
    """hipGraph capture/replay of the whole step (device-resident batch cursor, Philox offset, Adam bias terms) must
    reproduce the eager loop: same batches, same negatives, same weights."""
    from pykg2vec_amd.trainer import Trainer
    c = Case(name)
    out = []
    for use_graph in (False, True):
        # 400 train triples / 16 = 25 steps per epoch: one eager step, then multi-step graphs (GRAPH_UNROLL) and single-step
        # graphs of both parities all get replayed
        cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test, optimizer=opt, lr=0.02, batch_size=16)
        m = hip.model_from_case(c)
        tr = Trainer(m, cfg, use_graph=use_graph)
        tr.build_model()
        tr.generator = tr._new_generator()
        losses = [tr.train_model_epoch(e) for e in range(3)]
        assert (tr._graph is not None) == use_graph
        if use_graph:
            assert tr._graph_multi is not None
        out.append((losses, {k: p.detach().cpu().numpy() for k, p in hip.table_parameters(m)}))
    (l0, p0), (l1, p1) = out
    assert np.allclose(l0, l1, rtol=2e-4), (l0, l1)
    for k in p0:  # float atomics make the two runs differ in summation order only
        assert np.allclose(p0[k], p1[k], atol=2e-4, rtol=1e-3), (k, np.abs(p0[k] - p1[k]).max())


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("opt", ["sgd", "adam", "adagrad"])
def test_rescal_renormalisation_inside_the_optimiser_equals_the_separate_pass(hip, monkeypatch, opt, use_graph):
    """RESCAL renormalises its tables at the start of every forward (pairwise.py:843-844).  Inside an epoch the optimiser
    launch stores the rows already renormalised (kge_optimizer_step_rows) and the next step skips the pass; the epoch's last
    step does not, so the tables end the epoch exactly as with the separate pass (KGE_RESCAL_FUSED=0): not normalised."""
    from pykg2vec_amd.trainer import Trainer
    c = Case("rescal")
    out = []
    for fused in ("0", "1"):
        monkeypatch.setenv("KGE_RESCAL_FUSED", fused)
        cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test, optimizer=opt, lr=0.02, batch_size=16)
        m = hip.model_from_case(c)
        tr = Trainer(m, cfg, use_graph=use_graph)
        tr.build_model()
        tr.generator = tr._new_generator()
        losses = [tr.train_model_epoch(e) for e in range(3)]
        assert (tr._graph is not None) == use_graph
        out.append((losses, {k: p.detach().cpu().numpy() for k, p in hip.table_parameters(m)}))
    (l0, p0), (l1, p1) = out
    assert np.allclose(l0, l1, rtol=2e-4), (l0, l1)
    for k in p0:  # float atomics in the gradient: summation order only
        assert np.allclose(p0[k], p1[k], atol=2e-4, rtol=1e-3), (k, np.abs(p0[k] - p1[k]).max())
    norms = np.linalg.norm(p1["ent_embeddings.weight"], axis=1)
    assert np.abs(norms - 1.0).max() > 1e-4     # the last step's update was NOT followed by a renormalisation


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("opt", ["adam", "adagrad"])
def test_rescal_optimiser_reads_only_touched_gradient_rows(hip, monkeypatch, opt, use_graph):
    """With the step in one launch (k_rescal_pair) the entity rows that receive a gradient are marked in a bitmap and the
    row-owner optimiser skips the gradient read of every other row (two bitmaps alternating with the step parity, the
    optimiser resetting the other one).  Three epochs of eleven steps against the path without the bitmaps."""
    from pykg2vec_amd.trainer import Trainer
    E, R, k, B = 3000, 40, 64, 256
    rng = np.random.default_rng(5)
    train = np.stack([rng.integers(E, size=11 * B + 7), rng.integers(R, size=11 * B + 7), rng.integers(E, size=11 * B + 7)], 1)
    P = ko.init_params("rescal", rng, tot_entity=E, tot_relation=R, hidden_size=k)
    hp = dict(hidden_size=k, margin=1.0, neg_rate=1)
    out = []
    for fused in ("0", "1"):
        monkeypatch.setenv("KGE_RESCAL_FUSED", fused)
        cfg = hip.make_config(E, R, hp, train, train[:4], train[:4], optimizer=opt, lr=0.01, batch_size=B)
        m = hip.model_from_params("rescal", P, hp, E, R, train=train)
        tr = Trainer(m, cfg, use_graph=use_graph)
        tr.build_model()
        tr.generator = tr._new_generator()
        losses = [tr.train_model_epoch(e) for e in range(3)]
        assert (tr._touched is not None) == (fused == "1")
        if fused == "1":   # exactly one bitmap holds the last step's rows, the other was reset by its optimiser
            used = [int((b != 0).sum()) for b in tr._touched]
            assert min(used) == 0 and max(used) > 0, used
            assert bool((tr.flat.grad == 0).all())
        out.append((losses, {n: p.detach().cpu().numpy() for n, p in hip.table_parameters(m)}))
    (l0, p0), (l1, p1) = out
    assert np.allclose(l0, l1, rtol=2e-4), (l0, l1)
    for n in p0:
        assert np.allclose(p0[n], p1[n], atol=2e-4, rtol=1e-3), (n, np.abs(p0[n] - p1[n]).max())


@pytest.mark.parametrize("name", ["transe_l1", "transe_l2", "transh_l1", "transh_l2", "transd_l1", "transd_l2",
                                  "transm_l1", "transm_l2"])
def test_fused_sampler_step_equals_sample_then_step(hip, name):
    """kge_train_pairwise_hinge_sampled (corruption fused into the scoring kernel) must see exactly the batch
    kge_sample_batch emits for the same (start, n, seed, offset): same loss, same gradients."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.trainer import Trainer
    c = Case(name)
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test, batch_size=64)
    res = []
    for fused in (False, True):
        m = hip.model_from_case(c)
        tr = Trainer(m, cfg)
        tr.build_model()
        gen = tr._new_generator()
        tr.generator = gen
        tr.loss_buf.zero_()
        if fused:
            K.train_pairwise_hinge_sampled(tr._desc, gen.triples, gen.perm, 128, 64, None, gen.slots, 11, 999, 1.0, tr.loss_buf)
        else:
            b = K.sample_batch(gen.triples, gen.perm, 128, 64, 1, c.E, None, gen.slots, 11, 999)
            K.train_pairwise_hinge(tr._desc, *b, 1.0, tr.loss_buf)
        res.append((K.read_loss(tr.loss_buf).item(), [g.cpu().numpy().copy() for g in tr.flat.grad_views]))
    assert np.isclose(res[0][0], res[1][0], rtol=1e-5)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.allclose(a, b, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("name", ["distmult", "complex", "analogy", "cp", "simple", "simple_ignr", "quate"])
def test_pointwise_bundle_kernel_equals_row_kernel(hip, name):
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.trainer import Trainer
    c = Case(name)
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    res = []
    for bundle in (1, 2, 5):
        m = hip.model_from_case(c)
        tr = Trainer(m, cfg)
        tr.build_model()
        b = [hip.dev(x) for x in c.batch(0)]
        tr.loss_buf.zero_()
        K.train_pointwise_logistic(tr._desc, *b, m.kernel_lmbda(), m.kernel_reg_type(), tr.loss_buf, bundle=bundle)
        res.append((K.read_loss(tr.loss_buf).item(), [g.cpu().numpy().copy() for g in tr.flat.grad_views]))
    for other in res[1:]:
        assert np.isclose(res[0][0], other[0], rtol=1e-5)
        for a, b2 in zip(res[0][1], other[1]):
            assert np.allclose(a, b2, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("d,neg", [(40, 4), (100, 16), (1000, 16), (300, 7)])
def test_rotate_fused_sampler_bundle_equals_sample_then_step(hip, d, neg):
    """kge_train_pairwise_selfadv_sampled (RotatE: sampler fused, positive's rows and sin/cos kept in registers across
    the bundle) against kge_sample_batch + kge_train_pairwise_selfadv on the same Philox stream."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.trainer import Trainer
    rng = np.random.default_rng(d)
    E, R, B = 400, 9, 96
    hp = dict(hidden_size=d, margin=6.0, neg_rate=neg, alpha=0.8)
    P = ko.init_params("rotate", rng, tot_entity=E, tot_relation=R, hidden_size=d, margin=6.0)
    train = np.stack([rng.integers(E, size=1000), rng.integers(R, size=1000), rng.integers(E, size=1000)], 1)
    cfg = hip.make_config(E, R, hp, train, train[:4], train[:4], batch_size=B)
    res = []
    for fused in (False, True):
        m = hip.model_from_params("rotate", P, hp, E, R)
        tr = Trainer(m, cfg, use_graph=False)
        tr.build_model()
        gen = tr._new_generator()
        tr.generator = gen
        tr.loss_buf.zero_()
        if fused:
            K.train_pairwise_selfadv_sampled(tr._desc, gen.triples, gen.perm, 192, B, neg, 0.8, None, gen.slots, 3, 777, tr.loss_buf)
        else:
            b = K.sample_batch(gen.triples, gen.perm, 192, B, neg, E, None, gen.slots, 3, 777)
            K.train_pairwise_selfadv(tr._desc, *b, neg, 0.8, tr.loss_buf)
        res.append((K.read_loss(tr.loss_buf).item(), [g.cpu().numpy().copy() for g in tr.flat.grad_views]))
    assert np.isclose(res[0][0], res[1][0], rtol=2e-5), (res[0][0], res[1][0])
    for a, b2 in zip(res[0][1], res[1][1]):
        assert np.allclose(a, b2, atol=2e-6, rtol=2e-4), np.abs(a - b2).max()


@pytest.mark.parametrize("d,kr,E", [(40, 33, 300), (100, 100, 517), (8, 4, 65)])
def test_ntn_mfma_sweep_equals_batch_scorer_sweep(hip, d, kr, E):
    """The pre-contracted NTN sweep (one [E,d]x[d,k_r] f32-MFMA GEMM per query) against ranking every candidate triple
    through the batch scorer as the reference does (utils/evaluator.py:254-272)."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import build_filter_csr
    rng = np.random.default_rng(E)
    R = 7
    hp = dict(ent_hidden_size=d, rel_hidden_size=kr, lmbda=0.1, margin=1.0)
    P = ko.init_params("ntn", rng, tot_entity=E, tot_relation=R, ent_hidden_size=d, rel_hidden_size=kr)
    m = hip.model_from_params("ntn", P, hp, E, R)
    n = 300  # > one chunk of 256 triples
    test = np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1)
    hr_t, tr_h = ko.build_filters(test)
    csr = [hip.dev(a, torch.int64 if i % 2 == 0 else torch.int32) for i, a in enumerate(build_filter_csr(test, hr_t, tr_h))]
    trip = hip.dev(test)
    desc = m.make_desc()
    fast = K.eval_ranks(desc, trip, *csr).cpu().numpy()
    slow = K.eval_ranks_via_forward(desc, trip, *csr).cpu().numpy()
    assert (fast != slow).sum() <= 4 and np.abs(fast - slow).max() <= 1, (fast != slow).sum()
    sw = K.eval_sweep_scores(desc, trip[:3].contiguous()).cpu().numpy()
    for i in range(3):
        h, r, t = map(int, test[i])
        assert np.allclose(sw[2 * i], ko.sweep_scores("ntn", P, h, r, t, "tail"), atol=2e-5, rtol=2e-5)
        assert np.allclose(sw[2 * i + 1], ko.sweep_scores("ntn", P, h, r, t, "head"), atol=2e-5, rtol=2e-5)


def test_inference_hooks_and_state_dict_roundtrip(hip):
    """Evaluator.test_tail_rank / test_head_rank / test_rel_rank -- what the reference's Trainer.infer_tails / infer_heads /
    infer_rels call (utils/trainer.py:330-387) -- ride on the sweep hooks and the batch scorer; loading a state_dict
    (the reference's checkpoint format, same keys) keeps the tables in the trainer's flat buffer."""
    from pykg2vec_amd.trainer import Trainer
    c = Case("transe_l1")
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    m = hip.model_from_case(c, "adam.final.")
    tr = Trainer(m, cfg)
    tr.build_model()
    ev = tr.evaluator
    P = c.params("adam.final.")
    h, r, t = (int(x) for x in c.test[0])
    tails = ev.test_tail_rank(h, r, topk=5).cpu().numpy()
    want = np.argsort(-ko.sweep_scores("transe", P, h, r, t, "tail", **c.hp), kind="stable")[:5]
    assert list(tails) == [int(x) for x in want]
    heads = ev.test_head_rank(r, t, topk=3).cpu().numpy()
    want = np.argsort(-ko.sweep_scores("transe", P, h, r, t, "head", **c.hp), kind="stable")[:3]
    assert list(heads) == [int(x) for x in want]
    rels = ev.test_rel_rank(h, t, topk=c.R).cpu().numpy()
    ref = ko.score("transe", P, np.full(c.R, h), np.arange(c.R), np.full(c.R, t), **c.hp)
    assert list(rels) == [int(x) for x in np.argsort(-ref, kind="stable")]
    before = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        for p_ in m.parameters():
            p_.zero_()
    m.load_state_dict(before)
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), before[k])
    assert m.ent_embeddings.weight.data_ptr() == tr.flat.views[0].data_ptr()  # still backed by the flat buffer


def _csr_of(dense):
    off = np.concatenate([[0], np.cumsum(dense.sum(1).astype(np.int64))])
    ids = np.concatenate([np.flatnonzero(r) for r in dense]).astype(np.int32)
    return off, ids


@pytest.mark.parametrize("mode,ls", [("smooth", 0.1), ("plain", None)])
def test_head_1n_matches_reference_golden(hip, mode, ls):
    """1-N scoring head (MFMA GEMM + sigmoid) through autograd with Criterion.multi_class_bce, and the fused
    head + loss + backward entry point, against the live reference's numbers (tests/golden/ref_head_1n.npz)."""
    import os
    from golden_util import GOLDEN
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.criterion import Criterion
    from pykg2vec_amd.head import multi_class_bce_step, one_to_n_scores
    z = np.load(os.path.join(GOLDEN, "ref_head_1n.npz"))
    E = int(z["E"])
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    ent, bias = f(z["ent"]).requires_grad_(), f(z["bias"]).requires_grad_()
    xt, xh = f(z["x_t"]).requires_grad_(), f(z["x_h"]).requires_grad_()
    pt, ph = one_to_n_scores(xt, ent, bias), one_to_n_scores(xh, ent, bias)
    assert close(pt.detach().cpu().numpy(), z[mode + ".pred_t"]) and close(ph.detach().cpu().numpy(), z[mode + ".pred_h"])
    loss = Criterion.multi_class_bce(ph, pt, f(z["tr_h"]), f(z["hr_t"]), ls, E if ls is not None else None)
    loss.backward()
    assert close(loss.item(), z[mode + ".loss"])
    tol = dict(atol=2e-8, rtol=2e-4)  # gradients are O(1e-5): 1/(B*E) scaled
    for got, key in ((ent.grad, "g_ent"), (bias.grad, "g_bias"), (xt.grad, "g_x_t"), (xh.grad, "g_x_h")):
        assert np.allclose(got.cpu().numpy(), z[mode + "." + key], **tol), (key, np.abs(got.cpu().numpy() - z[mode + "." + key]).max())
    # fused form: two directions accumulate into the same loss / entity / bias gradient buffers
    loss_buf = K.new_loss_buffer("cuda")
    g_ent, g_bias = torch.zeros_like(ent), torch.zeros(E, device="cuda")
    dxs = []
    for x, lab in ((xt, z["hr_t"]), (xh, z["tr_h"])):
        off, ids = _csr_of(lab)
        dxs.append(multi_class_bce_step(x.detach(), ent.detach(), bias.detach(), torch.from_numpy(off).cuda(),
                                        torch.from_numpy(ids).cuda(), ls, loss_buf, g_ent, g_bias))
    assert close(K.read_loss(loss_buf).item(), z[mode + ".loss"])
    assert np.allclose(g_ent.cpu().numpy(), z[mode + ".g_ent"], **tol)
    assert np.allclose(g_bias.cpu().numpy().reshape(1, -1), z[mode + ".g_bias"], **tol)
    assert np.allclose(dxs[0].cpu().numpy(), z[mode + ".g_x_t"], **tol) and np.allclose(dxs[1].cpu().numpy(), z[mode + ".g_x_h"], **tol)


@pytest.mark.parametrize("B,E,d,with_bias", [(128, 14951, 200, True), (1, 65, 1, False), (1000, 4099, 33, True), (64, 64, 64, False)])
def test_head_1n_vs_oracle_other_shapes(hip, B, E, d, with_bias):
    from pykg2vec_amd import kernels as K
    rng = np.random.default_rng(B + E + d)
    x = rng.normal(size=(B, d)).astype(np.float32)
    ent = (rng.normal(size=(E, d)) * 0.2).astype(np.float32)
    bias = (rng.normal(size=E) * 0.1).astype(np.float32) if with_bias else None
    lab = (rng.random((B, E)) < 0.01).astype(np.float32)
    p_ref = ko.head_1n_forward(x, ent, bias)
    loss_ref, dp = ko.multi_class_bce_dir(p_ref, lab, 0.1, E)
    dx_ref, ge_ref, gb_ref = ko.head_1n_backward(x, ent, p_ref, dp)
    xd, ed = torch.from_numpy(x).cuda(), torch.from_numpy(ent).cuda()
    bd = torch.from_numpy(bias).cuda() if with_bias else None
    p = K.head_1n_forward(xd, ed, bd)
    assert close(p.cpu().numpy(), p_ref, atol=2e-6)
    dx, ge, gb = K.head_1n_backward(xd, ed, p, torch.from_numpy(dp).cuda(), need_bias=with_bias)
    scale = 1.0 / (B * E)
    assert np.allclose(dx.cpu().numpy(), dx_ref, atol=1e-3 * scale, rtol=1e-3)
    assert np.allclose(ge.cpu().numpy(), ge_ref, atol=1e-3 * scale, rtol=1e-3)
    if with_bias:
        assert np.allclose(gb.cpu().numpy(), gb_ref, atol=1e-3 * scale, rtol=1e-3)
    off, ids = _csr_of(lab)
    loss_buf = K.new_loss_buffer("cuda")
    g_ent = torch.zeros_like(ed)
    g_bias = torch.zeros(E, device="cuda") if with_bias else None
    dx2 = K.head_1n_bce(xd, ed, bd, torch.from_numpy(off).cuda(), torch.from_numpy(ids).cuda(), 0.1, loss_buf, g_ent, g_bias)
    assert np.isclose(K.read_loss(loss_buf).item(), loss_ref, rtol=2e-5)
    assert np.allclose(dx2.cpu().numpy(), dx_ref, atol=1e-3 * scale, rtol=1e-3)
    assert np.allclose(g_ent.cpu().numpy(), ge_ref, atol=1e-3 * scale, rtol=1e-3)


def _head_csr(labels):
    """CSR (int64 offsets, int32 ids) of the positive columns of a multi-hot [B, E] array."""
    off = np.zeros(labels.shape[0] + 1, np.int64)
    off[1:] = np.cumsum((labels > 0).sum(1))
    ids = np.concatenate([np.flatnonzero(row > 0) for row in labels]).astype(np.int32) if off[-1] else np.zeros(0, np.int32)
    return off, ids


def test_head_1n_rank_matches_reference_golden(hip):
    """kge_head_1n_rank against the LIVE REFERENCE's head outputs (tests/golden/ref_head_1n.npz: sigmoid(x @ ent.T + b) of
    projection.py:100-102): what the reference's evaluation makes of such a row -- topk(-preds) scanned from the best end by
    MetricCalculator.get_tail_rank (projection.py:119-125, utils/evaluator.py:70-123) -- is the number of entities predicted strictly
    above the true one, filtered by the row's known entities (the fixture's multi-hot hr_t / tr_h rows)."""
    import os
    from golden_util import GOLDEN
    from pykg2vec_amd.head import one_to_n_rank
    z = np.load(os.path.join(GOLDEN, "ref_head_1n.npz"))
    rng = np.random.default_rng(5)
    B, E = int(z["B"]), int(z["E"])
    for xs, preds, labels in ((z["x_t"], z["plain.pred_t"], z["hr_t"]), (z["x_h"], z["plain.pred_h"], z["tr_h"])):
        truth = rng.integers(E, size=B)
        off, ids = _head_csr(labels)
        got = one_to_n_rank(hip.dev(xs, torch.float32), hip.dev(z["ent"], torch.float32), hip.dev(z["bias"], torch.float32),
                            hip.dev(truth), hip.dev(off), hip.dev(ids, torch.int32)).cpu().numpy()
        for i in range(B):
            known = set(np.flatnonzero(labels[i] > 0).tolist())
            want = ko.rank_from_scores(-preds[i], int(truth[i]), known)
            if (int(got[0, i]), int(got[1, i])) != want:        # only a candidate within fp32 noise of the true one may move a rank
                near = int((np.abs(preds[i] - preds[i, truth[i]]) <= 2e-6).sum()) - 1
                assert abs(int(got[0, i]) - want[0]) <= near and abs(int(got[1, i]) - want[1]) <= near, (i, got[:, i].tolist(), want, near)


@pytest.mark.parametrize("B,E,d,with_bias", [(37, 203, 45, True), (5, 64, 8, False), (600, 5003, 200, True), (1030, 777, 33, False)])
def test_head_1n_rank_is_the_rank_of_its_own_energies(hip, B, E, d, with_bias):
    """Ranks are exact functions of the sweep's fp32 energies (the materialised form of the same call), the energies are the head's
    predictions negated, and nothing [B, E]-sized is needed for the ranks.  B >= 512 takes the matrix-core sweep."""
    from pykg2vec_amd import kernels as K
    rng = np.random.default_rng(B + E)
    x = rng.normal(size=(B, d)).astype(np.float32)
    ent = (rng.normal(size=(E, d)) * 0.3).astype(np.float32)
    bias = (rng.normal(size=E) * 0.1).astype(np.float32) if with_bias else None
    truth = rng.integers(E, size=B)
    labels = (rng.random((B, E)) < 0.02).astype(np.float32)
    labels[np.arange(B), truth] = 1.0
    labels[0] = 0.0                                   # a row without known entities
    off, ids = _head_csr(labels)
    xd, ed = hip.dev(x, torch.float32), hip.dev(ent, torch.float32)
    bd = hip.dev(bias, torch.float32) if with_bias else None
    ranks, ties = K.head_1n_rank(xd, ed, bd, hip.dev(truth), hip.dev(off), hip.dev(ids, torch.int32), return_ties=True)
    ranks, ties = ranks.cpu().numpy(), ties.cpu().numpy()
    en = K.head_1n_rank(xd, ed, bd, hip.dev(truth), energies=True).cpu().numpy()
    preds = K.head_1n_forward(xd, ed, bd).cpu().numpy()
    # (the rank sweep restarts its k chain every 64 elements from 512 rows on, the head's forward GEMM does not: logits differ by ~1e-6)
    assert np.allclose(-en, preds, atol=4e-6, rtol=1e-5), np.abs(en + preds).max()
    for i in range(B):
        known = set(np.flatnonzero(labels[i] > 0).tolist())
        assert (int(ranks[0, i]), int(ranks[1, i])) == ko.rank_from_scores(en[i], int(truth[i]), known), i
        assert ties[i] == int((en[i] == en[i, truth[i]]).sum()) - 1
    unf = K.head_1n_rank(xd, ed, bd, hip.dev(truth)).cpu().numpy()
    assert np.array_equal(unf[0], ranks[0]) and np.array_equal(unf[1], unf[0])       # no filter lists: filtered == raw


def test_head_1n_rank_reports_saturated_ties(hip):
    """sigmoid saturates to exactly 1.0f for logits beyond ~17: such candidates tie with a saturated true entity (the reference's
    order among them is torch.topk's, unspecified); the strict count leaves them out and the tie report names them."""
    from pykg2vec_amd import kernels as K
    E, d = 300, 16
    ent = np.zeros((E, d), np.float32)
    ent[:40, 0] = 30.0                                 # 40 saturated candidates for a query pointing along axis 0
    ent[40:, 0] = np.linspace(-3, 3, E - 40)
    x = np.zeros((2, d), np.float32)
    x[:, 0] = 1.0
    truth = np.asarray([3, 100])
    ranks, ties = K.head_1n_rank(hip.dev(x, torch.float32), hip.dev(ent, torch.float32), None, hip.dev(truth), return_ties=True)
    assert ranks[0].tolist() == [0, 40 + (E - 40 - 1 - 60)] and ties.tolist() == [39, 0]


@pytest.mark.parametrize("B,E,d,with_bias", [(128, 1000, 200, True), (70, 333, 50, False), (1, 64, 8, True), (257, 129, 97, True),
                                             (512, 14951, 200, True)])
def test_head_1n_bf16_option(hip, B, E, d, with_bias):
    """kge_head_1n_forward_bf16: operands rounded to bfloat16 (RNE), exact products, fp32 accumulation on v_mfma_f32_32x32x16_bf16.
    (i) Against the restatement with bf16-rounded operands: the logits agree to fp32 summation noise -- this also pins the MFMA
    operand layout (asymmetric random operands, ragged tile edges).  (ii) Against the fp32 head (what the reference computes,
    projection.py:100-102): within the bf16 operand rounding, |delta logit| <= 2^-7 * sum_k |x_k e_k| (each operand carries a
    relative rounding error of at most 2^-9).  The backward of one_to_n_scores(precision="bf16") is the fp32 backward on the saved
    predictions."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.head import one_to_n_scores
    rng = np.random.default_rng(B * 7 + E + d)
    x = rng.normal(size=(B, d)).astype(np.float32)
    ent = (rng.normal(size=(E, d)) * 0.3).astype(np.float32)
    bias = (rng.normal(size=E) * 0.1).astype(np.float32) if with_bias else None
    xd, ed = torch.from_numpy(x).cuda(), torch.from_numpy(ent).cuda()
    bd = torch.from_numpy(bias).cuda() if with_bias else None
    got = K.head_1n_forward(xd, ed, bd, precision="bf16").cpu().numpy()
    ref_bf16 = ko.head_1n_forward_bf16(x, ent, bias)
    logit = lambda p: np.log(p) - np.log1p(-p)
    ok = (ref_bf16 > 1e-3) & (ref_bf16 < 1 - 1e-3)      # where a float32 probability still resolves its logit
    scale = (np.abs(ko.bf16_round(x)) @ np.abs(ko.bf16_round(ent)).T)
    assert np.all(np.abs(logit(got[ok]) - logit(ref_bf16[ok])) <= 3e-6 * scale[ok] + 3e-4), \
        float(np.abs(logit(got[ok]) - logit(ref_bf16[ok])).max())
    assert np.allclose(got, ref_bf16, atol=2e-6, rtol=2e-5)
    ref_f32 = ko.head_1n_forward(x, ent, bias)
    okf = ok & (ref_f32 > 1e-3) & (ref_f32 < 1 - 1e-3)
    bound = 2.0 ** -7 * (np.abs(x) @ np.abs(ent).T) + 1e-4
    assert np.all(np.abs(logit(got[okf]) - logit(ref_f32[okf])) <= bound[okf])
    for tile in ("0", "1"):       # both tile shapes accumulate k in the same order: identical predictions
        K.set_switch("HEAD_TILE", int(tile))
        try:
            assert torch.equal(K.head_1n_forward(xd, ed, bd, precision="bf16").cpu(), torch.from_numpy(got)) or d % 4
        finally:
            K.set_switch("HEAD_TILE", None)
    # autograd form: forward in bf16, backward = the fp32 backward on the saved predictions
    xt = xd.clone().requires_grad_(True)
    et = ed.clone().requires_grad_(True)
    p16 = one_to_n_scores(xt, et, bd, precision="bf16")
    assert torch.equal(p16.detach().cpu(), torch.from_numpy(got))
    dp = torch.from_numpy(rng.normal(size=(B, E)).astype(np.float32)).cuda()
    p16.backward(dp)
    dx, ge, _ = K.head_1n_backward(xd, ed, p16.detach(), dp, need_bias=False)
    # (the backward GEMMs accumulate their split-K partial tiles with float atomics: equal to rounding, not bit for bit)
    assert torch.allclose(xt.grad, dx, rtol=1e-4, atol=1e-5) and torch.allclose(et.grad, ge, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,E,d", [(300, 1000, 40), (129, 257, 4), (2048, 14951, 200)])
def test_head_1n_large_tile_equals_small_tile(hip, monkeypatch, B, E, d):
    """The 128 x 128 macro-tile GEMM (large batches) keeps the k order of the 64 x 64 one: identical predictions, and the
    fused head + BCE entry point built on it agrees with the oracle."""
    from pykg2vec_amd import kernels as K
    rng = np.random.default_rng(B + E + d)
    x = rng.normal(size=(B, d)).astype(np.float32)
    ent = (rng.normal(size=(E, d)) * 0.2).astype(np.float32)
    bias = (rng.normal(size=E) * 0.1).astype(np.float32)
    xd, ed, bd = torch.from_numpy(x).cuda(), torch.from_numpy(ent).cuda(), torch.from_numpy(bias).cuda()
    out = {}
    for tile in ("0", "1"):
        monkeypatch.setenv("KGE_HEAD_TILE", tile)
        out[tile] = K.head_1n_forward(xd, ed, bd)
    assert torch.equal(out["0"], out["1"])
    p_ref = ko.head_1n_forward(x, ent, bias)
    assert close(out["1"].cpu().numpy(), p_ref, atol=2e-6)
    lab = (rng.random((B, E)) < 0.01).astype(np.float32)
    loss_ref, dp = ko.multi_class_bce_dir(p_ref, lab, 0.1, E)
    dx_ref, ge_ref, _ = ko.head_1n_backward(x, ent, p_ref, dp)
    off, ids = _csr_of(lab)
    loss_buf = K.new_loss_buffer("cuda")
    g_ent = torch.zeros_like(ed)
    dx = K.head_1n_bce(xd, ed, bd, torch.from_numpy(off).cuda(), torch.from_numpy(ids).cuda(), 0.1, loss_buf, g_ent, torch.zeros(E, device="cuda"))
    scale = 1.0 / (B * E)
    assert np.isclose(K.read_loss(loss_buf).item(), loss_ref, rtol=2e-5)
    assert np.allclose(dx.cpu().numpy(), dx_ref, atol=1e-3 * scale, rtol=1e-3)
    assert np.allclose(g_ent.cpu().numpy(), ge_ref, atol=1e-3 * scale, rtol=1e-3)


@pytest.mark.parametrize("B,E,d", [(130, 259, 8), (257, 129, 12), (64, 1001, 200), (700, 14951, 200)])
def test_head_1n_wide_backward_products(hip, B, E, d):
    """The 128 x 128 forms of the two backward products (dX = dZ Ent, g_ent += dZ^T X, g_bias += column sums), forced on at ragged
    shapes: the autograd form reads dpreds * p (1 - p) through 4-byte aligned 16-byte loads (E odd: rows are not 16-byte aligned, the
    tail of the last row is fetched from the last in-bounds position), the fused form reads the padded workspace rows.  Both against
    the oracle, and both bit-reproducible run to run (split-K partial tiles are summed in split order, no float atomics)."""
    from pykg2vec_amd import kernels as K
    rng = np.random.default_rng(B * 3 + E + d)
    x = rng.normal(size=(B, d)).astype(np.float32)
    ent = (rng.normal(size=(E, d)) * 0.2).astype(np.float32)
    bias = (rng.normal(size=E) * 0.1).astype(np.float32)
    lab = (rng.random((B, E)) < 0.01).astype(np.float32)
    p_ref = ko.head_1n_forward(x, ent, bias)
    loss_ref, dp = ko.multi_class_bce_dir(p_ref, lab, 0.1, E)
    dx_ref, ge_ref, gb_ref = ko.head_1n_backward(x, ent, p_ref, dp)
    xd, ed, bd = torch.from_numpy(x).cuda(), torch.from_numpy(ent).cuda(), torch.from_numpy(bias).cuda()
    off, ids = _csr_of(lab)
    offd, idsd = torch.from_numpy(off).cuda(), torch.from_numpy(ids).cuda()
    scale = 1.0 / (B * E)
    K.set_switch("HEAD_TILE", 1)
    try:
        p = K.head_1n_forward(xd, ed, bd)
        runs = []
        for _ in range(2):
            dx, ge, gb = K.head_1n_backward(xd, ed, p, torch.from_numpy(dp).cuda())
            loss_buf = K.new_loss_buffer("cuda")
            g_ent, g_bias = torch.zeros_like(ed), torch.zeros(E, device="cuda")
            dx2 = K.head_1n_bce(xd, ed, bd, offd, idsd, 0.1, loss_buf, g_ent, g_bias)
            runs.append((dx, ge, gb, dx2, g_ent, g_bias))
    finally:
        K.set_switch("HEAD_TILE", None)
    for got, ref in zip(runs[0], (dx_ref, ge_ref, gb_ref, dx_ref, ge_ref, gb_ref)):
        assert np.allclose(got.cpu().numpy(), ref, atol=1e-3 * scale, rtol=1e-3)
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    # the workspace-free form of the entry point (partial tiles meet in float atomics), both tile shapes
    for tile in (0, 1):
        K.set_switch("HEAD_TILE", tile)
        try:
            got = K.head_1n_backward(xd, ed, p, torch.from_numpy(dp).cuda(), workspace=False)
        finally:
            K.set_switch("HEAD_TILE", None)
        for g, ref in zip(got, (dx_ref, ge_ref, gb_ref)):
            assert np.allclose(g.cpu().numpy(), ref, atol=1e-3 * scale, rtol=1e-3)


@pytest.mark.parametrize("name,neg", [("distmult", 1), ("complex", 3), ("analogy", 1), ("cp", 2), ("simple", 1), ("quate", 4)])
def test_fused_pointwise_sampler_step_equals_sample_then_step(hip, name, neg):
    """kge_train_pointwise_logistic_sampled (corruption fused into the pointwise kernel) must see exactly the rows
    kge_sample_batch(layout pointwise) emits for the same (start, n, neg_rate, seed, offset): same loss, same gradients."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.trainer import Trainer
    c = Case(name)
    cfg = hip.make_config(c.E, c.R, dict(c.hp, neg_rate=neg), c.train, c.valid, c.test, batch_size=64)
    res = []
    for fused in (False, True):
        m = hip.model_from_case(c)
        tr = Trainer(m, cfg)
        tr.build_model()
        gen = tr._new_generator()
        tr.generator = gen
        tr.loss_buf.zero_()
        if fused:
            K.train_pointwise_logistic_sampled(tr._desc, gen.triples, gen.perm, 128, 64, neg, None, gen.slots, 11, 999,
                                               m.kernel_lmbda(), m.kernel_reg_type(), tr.loss_buf)
        else:
            b = K.sample_batch(gen.triples, gen.perm, 128, 64, neg, c.E, None, gen.slots, 11, 999, pointwise=True)
            K.train_pointwise_logistic(tr._desc, *b, m.kernel_lmbda(), m.kernel_reg_type(), tr.loss_buf, bundle=1 + neg)
        res.append((K.read_loss(tr.loss_buf).item(), [g.cpu().numpy().copy() for g in tr.flat.grad_views]))
    assert np.isclose(res[0][0], res[1][0], rtol=1e-5)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.allclose(a, b, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("name", ["transe_l1", "transh_l2", "transd_l1", "transm_l1", "distmult", "complex", "rotate"])
def test_fused_sampler_steps_with_bern_probabilities(hip, name):
    """The sampler-fused kernels with per-relation bern head/tail probabilities (data/generator.py:77-83) must see the batch
    kge_sample_batch emits with the same probabilities."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.trainer import Trainer
    c = Case(name)
    neg = int(c.hp.get("neg_rate", 1)) if c.model == "rotate" else (2 if c.pointwise else 1)
    cfg = hip.make_config(c.E, c.R, dict(c.hp, neg_rate=neg), c.train, c.valid, c.test, batch_size=64)
    bern = torch.tensor(np.random.default_rng(5).uniform(0.05, 0.95, size=c.R), dtype=torch.float32, device="cuda")
    res = []
    for fused in (False, True):
        m = hip.model_from_case(c)
        tr = Trainer(m, cfg)
        tr.build_model()
        gen = tr._new_generator()
        tr.loss_buf.zero_()
        args = (gen.triples, gen.perm, 64, 64)
        if c.pointwise:
            if fused:
                K.train_pointwise_logistic_sampled(tr._desc, *args, neg, bern, gen.slots, 3, 77, m.kernel_lmbda(),
                                                   m.kernel_reg_type(), tr.loss_buf)
            else:
                b = K.sample_batch(*args, neg, c.E, bern, gen.slots, 3, 77, pointwise=True)
                K.train_pointwise_logistic(tr._desc, *b, m.kernel_lmbda(), m.kernel_reg_type(), tr.loss_buf, bundle=1 + neg)
        elif c.model == "rotate":
            if fused:
                K.train_pairwise_selfadv_sampled(tr._desc, *args, neg, c.hp["alpha"], bern, gen.slots, 3, 77, tr.loss_buf)
            else:
                b = K.sample_batch(*args, neg, c.E, bern, gen.slots, 3, 77)
                K.train_pairwise_selfadv(tr._desc, *b, neg, c.hp["alpha"], tr.loss_buf)
        else:
            if fused:
                K.train_pairwise_hinge_sampled(tr._desc, *args, bern, gen.slots, 3, 77, 1.0, tr.loss_buf)
            else:
                b = K.sample_batch(*args, 1, c.E, bern, gen.slots, 3, 77)
                K.train_pairwise_hinge(tr._desc, *b, 1.0, tr.loss_buf)
        res.append((K.read_loss(tr.loss_buf).item(), [g.cpu().numpy().copy() for g in tr.flat.grad_views]))
    assert np.isclose(res[0][0], res[1][0], rtol=1e-5)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.allclose(a, b, atol=1e-5, rtol=1e-4)


GEMM_CASES = ["distmult", "complex", "complexn3", "analogy", "rescal", "cp", "simple", "simple_ignr", "quate", "rotate"]


@pytest.mark.parametrize("name", GEMM_CASES)
def test_matrix_core_sweep_scores_and_ranks(hip, name, monkeypatch):
    """k_eval_gemm (the dot-product sweep on the f32 matrix cores), forced on for the small golden tables: energies within
    the fp32 tolerance of the live reference's, integer ranks EXACT functions of the kernel's own energies (the target
    energy comes from k_eval_target_filter_chain, which must round like the MFMA chain), and ranks against the
    reference inside the score band."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    monkeypatch.setenv("KGE_EVAL_GEMM", "1")
    c = Case(name)
    m = hip.model_from_case(c, "adam.final.")
    cfg = hip.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test)
    if c.model == "rescal":
        m.normalize_tables()
    sw = K.eval_sweep_scores(m.make_desc(), hip.dev(c.test[:4])).cpu().numpy()
    assert close(sw, c.z["eval.sweeps"], atol=2e-5, rtol=2e-5), np.abs(sw - c.z["eval.sweeps"]).max()
    n = len(c.z["eval.rank_head"])
    ranks = Evaluator(m, cfg).rank_all(c.test, n).cpu().numpy()
    scores = K.eval_sweep_scores(m.make_desc(), hip.dev(c.test[:n])).cpu().numpy()
    hr_t, tr_h = c.filters()
    for i, (h, r, t) in enumerate(c.test[:n]):
        rt = ko.rank_from_scores(scores[2 * i], int(t), hr_t[(int(h), int(r))])
        rh = ko.rank_from_scores(scores[2 * i + 1], int(h), tr_h[(int(t), int(r))])
        assert (ranks[1, i], ranks[3, i]) == rt and (ranks[0, i], ranks[2, i]) == rh, (i, ranks[:, i], rt, rh)
    ref = np.stack([c.z["eval.rank_head"], c.z["eval.rank_tail"], c.z["eval.frank_head"], c.z["eval.frank_tail"]])
    assert_ranks_inside_band(name + "_gemm", ranks, ref, scores, c.test[:n])


@pytest.mark.parametrize("model,d", [("distmult", 50), ("rotate", 50), ("complex", 52), ("rotate", 300), ("quate", 12)])
def test_matrix_core_sweep_ranks_with_long_filter_lists(hip, model, d, monkeypatch):
    """k_eval_target_filter_chain takes a query's (true candidate + known entities) list in groups of 4 / 16 / 64 pairs with
    K chunks of 256 / 64 / 16: lists of every length class (empty ... > 128 known entities, the true candidate inside the
    list), table widths that are not multiples of 4 (scalar re-layout path) and K that is no multiple of any chunk length.
    Ranks must be exact functions of the sweep's own energies (k_eval_gemm and the chain kernel round alike)."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    monkeypatch.setenv("KGE_EVAL_GEMM", "1")
    rng = np.random.default_rng(5)
    E, R, n = 300, 3, 280
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, hidden_size=d, margin=6.0)
    # known tails per (h, r): h = 0 -> 150, h = 1 -> 70, h = 2 -> 17, h = 3 -> 5, then a random background
    heavy = [(h, 0, t) for h, cnt in ((0, 150), (1, 70), (2, 17), (3, 5)) for t in rng.choice(E, size=cnt, replace=False)]
    back = np.stack([rng.integers(E, size=2500), rng.integers(R, size=2500), rng.integers(E, size=2500)], 1)
    known = np.concatenate([np.asarray(heavy, dtype=np.int64), back])
    test = np.concatenate([np.asarray(heavy, dtype=np.int64)[::9][:40], back[:n - 40]])[:n]
    hp = dict(hidden_size=d, lmbda=0.01, margin=6.0, alpha=1.0)
    cfg = hip.make_config(E, R, hp, known, known[:8], test)
    m = hip.model_from_params(model, P, hp, E, R)
    ranks = Evaluator(m, cfg).rank_all(test, n).cpu().numpy()
    scores = K.eval_sweep_scores(m.make_desc(), hip.dev(test)).cpu().numpy()
    hr_t, tr_h = cfg.knowledge_graph.cache["hr_t"], cfg.knowledge_graph.cache["tr_h"]
    assert max(len(v) for v in hr_t.values()) >= 150
    for i, (h, r, t) in enumerate(test):
        assert (ranks[1, i], ranks[3, i]) == ko.rank_from_scores(scores[2 * i], int(t), hr_t[(int(h), int(r))]), i
        assert (ranks[0, i], ranks[2, i]) == ko.rank_from_scores(scores[2 * i + 1], int(h), tr_h[(int(t), int(r))]), i


@pytest.mark.parametrize("model,E,d", [("distmult", 333, 100), ("complex", 1000, 36), ("distmult", 129, 8)])
def test_matrix_core_sweep_equals_vector_sweep_ranks_on_ragged_shapes(hip, model, E, d, monkeypatch):
    """Odd numbers of 64-candidate tiles, partial last tiles, query counts that are not multiples of 128: ranks from the
    matrix-core sweep are exact functions of its energies, and agree with the VALU sweep's up to fp32 near-ties."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    rng = np.random.default_rng(11)
    R, n = 5, 333
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, hidden_size=d)
    trip = np.stack([rng.integers(E, size=n + 600), rng.integers(R, size=n + 600), rng.integers(E, size=n + 600)], 1)
    hp = dict(hidden_size=d, lmbda=0.01)
    cfg = hip.make_config(E, R, hp, trip[n:], trip[:8], trip[:n])
    m = hip.model_from_params(model, P, hp, E, R)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("KGE_EVAL_GEMM", flag)
        out[flag] = (Evaluator(m, cfg).rank_all(trip[:n], n).cpu().numpy(),
                     K.eval_sweep_scores(m.make_desc(), hip.dev(trip[:n])).cpu().numpy())
    r1, s1 = out["1"]
    hr_t, tr_h = cfg.knowledge_graph.cache["hr_t"], cfg.knowledge_graph.cache["tr_h"]
    for i, (h, r, t) in enumerate(trip[:n]):
        assert (r1[1, i], r1[3, i]) == ko.rank_from_scores(s1[2 * i], int(t), hr_t[(int(h), int(r))])
        assert (r1[0, i], r1[2, i]) == ko.rank_from_scores(s1[2 * i + 1], int(h), tr_h[(int(t), int(r))])
    assert np.allclose(out["0"][1], s1, atol=1e-6, rtol=1e-5)
    assert (out["0"][0] != r1).mean() < 0.01 and np.abs(out["0"][0] - r1).max() <= 2


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipGraph"])
@pytest.mark.parametrize("opt", ["adam", "sgd", "adagrad"])
@pytest.mark.parametrize("k,R,B", [(64, 40, 256), (200, 7, 96), (52, 300, 512)])
def test_rescal_staged_entity_gradients_are_reproducible_and_equal_the_atomic_step(hip, monkeypatch, opt, use_graph, k, R, B):
    """Round 5: the entity gradients of the pairwise RESCAL step staged per (pair, side) and summed by the row owners in slot order
    (kge_rescal_pair_step_staged + kge_optimizer_step_rows_staged) instead of float atomics.  Two runs of the staged path must end
    with BYTE-IDENTICAL tables and optimiser state (deterministic grouping, no atomics on any gradient; no relation exceeds one
    64-pair chunk here), and agree with the atomic path to rounding."""
    from pykg2vec_amd.trainer import Trainer
    E = 3000
    rng = np.random.default_rng(11)
    n = 9 * B + 5
    train = np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1)
    train[: B // 2, 0] = 17          # a hub entity: more registrations than the bucket holds (overflow chain)
    train = train[rng.permutation(n)]
    P = ko.init_params("rescal", rng, tot_entity=E, tot_relation=R, hidden_size=k)
    hp = dict(hidden_size=k, margin=1.0, neg_rate=1)
    monkeypatch.setenv("KGE_RESCAL_FUSED", "1")
    out = []
    for staged in ("1", "1", "0"):
        monkeypatch.setenv("KGE_RESCAL_STAGED", staged)
        cfg = hip.make_config(E, R, hp, train, train[:4], train[:4], optimizer=opt, lr=0.01, batch_size=B)
        m = hip.model_from_params("rescal", P, hp, E, R, train=train)
        tr = Trainer(m, cfg, use_graph=use_graph)
        tr.build_model()
        tr.generator = tr._new_generator()
        losses = [tr.train_model_epoch(e) for e in range(3)]
        took = getattr(tr, "_rescal_stage", None) is not None
        assert took == (staged == "1" and k % 4 == 0), (took, staged)
        if took:   # which side of the 64-pairs-per-relation line the run is on is computed up front and reported
            worst = max(int(np.bincount(train[tr.generator.perm.cpu().numpy()[lo:lo + B], 1], minlength=R).max()) for lo in range(0, n, B))
            assert tr.rescal_reproducible == (worst <= 64), (tr.rescal_reproducible, worst)
        if took:   # every list the optimiser consumed was reset
            assert int(tr._rescal_stage.count.abs().sum()) == 0 and int(tr._rescal_stage.head.abs().sum()) == 0
            assert bool((tr.flat.grad[: E * k] == 0).all())
        out.append((losses, tr.flat.param.clone(), None if tr.flat.state1 is None else tr.flat.state1.clone()))
    (la, pa, sa), (lb, pb, sb), (lc, pc, sc) = out
    if max(np.bincount(train[:, 1], minlength=R)) * B // n <= 48:   # (relations comfortably inside one chunk per batch)
        assert torch.equal(pa, pb) and (sa is None or torch.equal(sa, sb))
    assert np.allclose(la, lc, rtol=5e-4), (la, lc)
    # against the atomic path: equal to rounding -- except that Adam / Adagrad turn a rounding-residue gradient into a full +-lr step, so
    # isolated entries may differ by a few lr (as in the pull-vs-push tests); the atomic path itself differs run to run by as much
    off = float((~torch.isclose(pa, pc, atol=2e-4, rtol=1e-3)).float().mean())
    assert off <= 2e-3, (off, float((pa - pc).abs().max()))


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipGraph"])
@pytest.mark.parametrize("opt,k", [("adam", 128), ("sgd", 128), ("adagrad", 128), ("rms", 128), ("adam", 130)])
def test_rescal_optimizer_rider_is_bit_identical_to_the_separate_launches(hip, monkeypatch, opt, k, use_graph):
    """Round 6: kge_optimizer_step_rows_rownorm -- the relation matrices' optimiser riding in the first workgroups of the entity table's
    row-owner sweep (k_opt_rows4<..., RIDER>) + the rescale launch -- must leave BYTE-IDENTICAL tables and optimiser state to
    kge_optimizer_step_rows(_staged) followed by kge_optimizer_step_rownorm (KGE_OPT_RIDER=0): same device functions, same order."""
    from pykg2vec_amd.trainer import Trainer
    E, R, B = 1500, 6, 96                  # k = 128: rel_matrices rows of 16 384 floats, the wide-row form's minimum; k = 130: rows that
                                           # are no multiple of 4 floats keep the separate launches (the rider form is float4-only)
    rng = np.random.default_rng(23)
    n = 7 * B
    train = np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1)
    P = ko.init_params("rescal", rng, tot_entity=E, tot_relation=R, hidden_size=k)
    hp = dict(hidden_size=k, margin=1.0, neg_rate=1)
    monkeypatch.setenv("KGE_RESCAL_FUSED", "1")
    out = []
    for rider in ("1", "0"):
        monkeypatch.setenv("KGE_OPT_RIDER", rider)
        cfg = hip.make_config(E, R, hp, train, train[:4], train[:4], optimizer=opt, lr=0.01, batch_size=B)
        tr = Trainer(hip.model_from_params("rescal", P, hp, E, R, train=train), cfg, use_graph=use_graph)
        tr.build_model()
        tr.generator = tr._new_generator()
        losses = [tr.train_model_epoch(e) for e in range(2)]
        out.append((losses, tr.flat.param.clone(), None if tr.flat.state1 is None else tr.flat.state1.clone(),
                    None if tr.flat.state2 is None else tr.flat.state2.clone()))
    (la, pa, sa, ta), (lb, pb, sb, tb) = out
    if getattr(tr, "rescal_reproducible", None) and getattr(tr, "_rescal_stage", None) is not None:   # (the staged pair step is run-to-run
                                                                                                      # reproducible on this graph: then so is the whole)
        assert np.allclose(la, lb, rtol=1e-6)      # (the epoch loss is summed over slots with float atomics: equal to rounding)
        assert torch.equal(pa, pb) and (sa is None or torch.equal(sa, sb)) and (ta is None or torch.equal(ta, tb))
    else:
        # (the atomic pair step differs run to run by rounding, and Adam / RMSprop turn a rounding-residue gradient into a full lr step:
        #  isolated entries -- the same bar as the staged-vs-atomic test above)
        assert np.allclose(la, lb, rtol=5e-4), (la, lb)
        off = float((~torch.isclose(pa, pb, atol=2e-4, rtol=1e-3)).float().mean())
        assert off <= 2e-3, (off, float((pa - pb).abs().max()))


def test_rescal_staged_lists_are_emptied_when_an_epoch_dies_between_pair_step_and_optimiser(hip, monkeypatch):
    """The staged RESCAL gradients live in per-entity lists that the pair step fills and the row-owner optimiser consumes and resets.
    An exception between the two (here: the optimiser call of the second batch raises) must not leave registrations behind:
    Trainer.train_model_epoch empties the lists and the touched-row bitmaps before re-raising, and the next epoch trains normally."""
    from pykg2vec_amd.trainer import Trainer
    E, R, k, B = 2000, 20, 64, 128
    rng = np.random.default_rng(5)
    n = 6 * B
    train = np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1)
    P = ko.init_params("rescal", rng, tot_entity=E, tot_relation=R, hidden_size=k)
    hp = dict(hidden_size=k, margin=1.0, neg_rate=1)
    monkeypatch.setenv("KGE_RESCAL_FUSED", "1")
    monkeypatch.setenv("KGE_RESCAL_STAGED", "1")
    cfg = hip.make_config(E, R, hp, train, train[:4], train[:4], optimizer="adam", lr=0.01, batch_size=B)
    tr = Trainer(hip.model_from_params("rescal", P, hp, E, R, train=train), cfg, use_graph=False)
    tr.build_model()
    tr.generator = tr._new_generator()
    first = tr.train_model_epoch(0)
    st = tr._rescal_stage
    assert st is not None and np.isfinite(first)
    real, calls = tr.flat.optimizer_step_rows_first, []

    def dies_on_second_call(*a, **kw):
        calls.append(1)
        if len(calls) == 2:
            raise RuntimeError("injected")
        return real(*a, **kw)
    monkeypatch.setattr(tr.flat, "optimizer_step_rows_first", dies_on_second_call)
    with pytest.raises(RuntimeError, match="injected"):
        tr.train_model_epoch(1)
    torch.cuda.synchronize()
    assert int(st.count.abs().sum()) == 0 and int(st.head.abs().sum()) == 0
    assert all(int(b.abs().sum()) == 0 for b in tr._touched)
    monkeypatch.setattr(tr.flat, "optimizer_step_rows_first", real)
    tr.flat.grad.zero_()      # (the half-finished step's relation gradient; entity gradients never reached the buffer)
    after = tr.train_model_epoch(2)
    assert np.isfinite(after) and after < first
    assert int(st.count.abs().sum()) == 0 and int(st.head.abs().sum()) == 0
    assert bool(torch.isfinite(tr.flat.param).all())


@pytest.mark.parametrize("opt", ["adam", "sgd", "adagrad", "rms"])
@pytest.mark.parametrize("rows,k", [(7, 136), (37, 200), (3, 128)])
def test_optimizer_rownorm_equals_sweep_plus_normalisation_bit_for_bit(hip, opt, rows, k):
    """kge_optimizer_step_rownorm (round 5): the dense optimiser over RESCAL's relation matrices with the rows' sums of squares left by
    the optimiser launch itself + one rescale launch must store exactly what kge_optimizer_step + kge_rescal_normalize_ws store (same
    chunking, same element-to-thread map and summation order), for several steps, with and without gradients."""
    from pykg2vec_amd import kernels as K
    dim = k * k
    assert K.optimizer_step_rownorm_ok(rows, dim)
    g0 = torch.Generator(device="cpu").manual_seed(rows * 1000 + k)
    p = torch.randn(rows * dim, generator=g0).cuda()
    runs = []
    for fused in (False, True):
        pa = p.clone()
        s1 = None if opt == "sgd" else torch.zeros_like(pa)
        s2 = torch.zeros_like(pa) if opt == "adam" else None
        for step in range(1, 5):
            g = (torch.randn(rows * dim, generator=torch.Generator(device="cpu").manual_seed(step)) * (step % 2)).cuda()   # zero on even steps
            if fused:
                K.optimizer_step_rownorm(opt, pa, g, s1, s2, rows, dim, 0.01, step)
            else:
                K.optimizer_step(opt, pa, g, s1, s2, 0.01, step)
                K.rescal_normalize_relations(pa.view(rows, dim), k)
            assert bool((g == 0).all())
        runs.append((pa, s1, s2))
    for a, b in zip(*runs):
        assert (a is None and b is None) or torch.equal(a, b)
    assert torch.allclose(runs[1][0].view(rows, dim).norm(dim=1), torch.ones(rows, device="cuda"), atol=1e-5)
