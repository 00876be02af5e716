"""Filtered ranks on TRAINED tables against the live reference's (round 6; fixtures: oracle/make_golden_trained.py).

  * the WHOLE pretrained FB15k TransE checkpoint of the reference (examples/pretrained/TransE/model.vec.pt: E = 14 951, d = 50), L1 and
    L2, 1 024 test triples -- the tables travel inside the fixture;
  * the BASELINE shapes C1 / C2 / C3 after golden_util.TRAINED[case] epochs of training: the tables are re-produced on the GPU by the
    drop-in Trainer's bit-reproducible default step path and recognised by their SHA-256 (a changed digest means the training
    arithmetic changed: regenerate with tools/make_trained_tables.py + oracle/make_golden_trained.py).

The bar: >= 99.5 % of the rank entries identical to the reference's -- or, where the reference's OWN fp32 ranks already differ from
the float64 ranks of the same tables on more than 0.2 % of the entries (C1 trained: 1.4 %), within 2.5x that noise: two independent
fp32 summation orders cannot agree more often than each agrees with the exact ranks --, the HIP ranks no further from float64 than
the reference's are (x 1.5 + 0.2 %), and every differing entry explained by candidates inside the fp32 band around the true
candidate's energy (golden_util.rank_band_ok); float64 is the arbiter in the report (gpurun_out/rank_agreement_trained.json ->
profiles/r06_rank_agreement_trained.json)."""
import json
import os

import numpy as np
import pytest

import golden_util as gu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAINED = list(gu.TRAINED)
MIN_IDENTICAL = 0.995
FLAT_BAR = ("pretrained_fb15k_l1", "pretrained_fb15k_l2", "trained_c1_transe_l1", "trained_c2_complex")


def _fixture(fname):
    path = os.path.join(GOLDEN, fname)
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated" % fname)
    return np.load(path)


def _fixture_sane(ref, r64, n):
    assert ref.shape == (4, n) and r64.shape == ref.shape
    assert (ref[2] <= ref[0]).all() and (ref[3] <= ref[1]).all()          # filtered <= raw
    assert (ref != r64).mean() <= 0.02 and np.abs(ref - r64).max() <= 3    # the reference's own fp32 noise on trained tables


def test_pretrained_fixture_is_the_whole_checkpoint():
    z = _fixture("ref_trained_ranks_pretrained_fb15k.npz")
    assert z["ent_embeddings"].shape == (14951, 50) and z["rel_embeddings"].shape == (1345, 50) and int(z["n"]) == 1024
    for tag in ("l1", "l2"):
        _fixture_sane(z["ranks_" + tag], z["ranks64_" + tag], 1024)
    # the plausible half ranks near the top under the norm the checkpoint was trained with
    assert np.median(z["ranks_l1"][3, 512:]) < 100 < np.median(z["ranks_l1"][3, :512])


@pytest.mark.parametrize("name", TRAINED)
def test_trained_fixture_is_consistent_with_float64(name):
    z = _fixture("ref_trained_ranks_%s.npz" % name)
    n = gu.TRAINED[name]["n_rank"]
    _fixture_sane(z["ranks"], z["ranks64"], n)
    assert float(z["last_loss"]) < float(z["first_loss"])                  # the tables were trained
    # training triples rank near the top of their sweeps, held-out (random) triples do not
    assert np.median(z["ranks"][3, n - n // 2:]) < np.median(z["ranks"][3, :n - n // 2])


def _compare(tag, m, cfg, queries, ref, r64, true_scores):
    """Evaluator.rank_all on the HIP path vs the reference's ranks; the report lists every differing entry with the float64 verdict."""
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    import hip_util
    n = len(queries)
    ranks = Evaluator(m, cfg).rank_all(queries, n).cpu().numpy()
    differ = np.flatnonzero((ranks != ref).any(0))
    report = {"case": tag, "test_triples": n, "rank_entries": int(ranks.size), "entries_differing": int((ranks != ref).sum()),
              "identical_fraction": float((ranks == ref).mean()), "triples_with_a_differing_rank": int(len(differ)),
              "hip_equals_float64": 0, "reference_equals_float64": 0, "neither": 0, "max_abs_rank_diff": 0,
              "reference_entries_differing_from_float64": int((ref != r64).sum()),
              "hip_entries_differing_from_float64": int((ranks != r64).sum()), "flips": []}
    if len(differ):
        scores = K.eval_sweep_scores(m.make_desc(), hip_util.dev(queries[differ])).cpu().numpy()
    for j, i in enumerate(differ):
        h, r, t = (int(x) for x in queries[i])
        for side, row, true, (a, b) in (("tail", scores[2 * j], t, (1, 3)), ("head", scores[2 * j + 1], h, (0, 2))):
            assert np.isclose(row[true], true_scores[i, 0 if side == "head" else 1], atol=2e-5, rtol=2e-5)
            for which in (a, b):
                g_, r_, d_ = int(ranks[which, i]), int(ref[which, i]), int(r64[which, i])
                if g_ == r_:
                    continue
                ok, near = gu.rank_band_ok(row, true, g_, r_)
                assert ok, (tag, int(i), side, which, g_, r_, near)
                report["max_abs_rank_diff"] = max(report["max_abs_rank_diff"], abs(g_ - r_))
                verdict = "hip" if g_ == d_ else ("reference" if r_ == d_ else "neither")
                report["hip_equals_float64" if verdict == "hip" else "reference_equals_float64" if verdict == "reference" else "neither"] += 1
                report["flips"].append({"triple": int(i), "side": side, "filtered": which >= 2, "hip": g_, "reference": r_, "float64": d_,
                                        "float64_sides_with": verdict, "candidates_inside_band": near})
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "rank_agreement_trained.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[tag] = report
    json.dump(doc, open(path, "w"), indent=1)
    ref_noise = report["reference_entries_differing_from_float64"] / ranks.size
    hip_noise = report["hip_entries_differing_from_float64"] / ranks.size
    # measured (profiles/r06_rank_agreement_trained.json): 99.51 / 99.71 % (pretrained L1 / L2), 99.51 % (C1), 99.80 % (C2) -- every
    # path involved is deterministic, so the flat 99.5 % bar holds exactly; a case whose reference is noisier than 0.2 % against
    # float64 AND has not been measured yet falls back to 2.5x that noise
    report["required_identical_fraction"] = MIN_IDENTICAL if tag in FLAT_BAR else min(MIN_IDENTICAL, 1.0 - 2.5 * ref_noise)
    json.dump(doc, open(path, "w"), indent=1)
    assert report["identical_fraction"] >= report["required_identical_fraction"], report
    assert hip_noise <= 1.5 * ref_noise + 0.002, report       # as close to the exact ranks as the reference is
    return report


@pytest.mark.gpu
@pytest.mark.parametrize("l1", [True, False])
def test_hip_ranks_on_the_whole_pretrained_fb15k_checkpoint(l1):
    import hip_util
    z = _fixture("ref_trained_ranks_pretrained_fb15k.npz")
    tag = "l1" if l1 else "l2"
    E, R = int(z["E"]), int(z["R"])
    hp = dict(hidden_size=50, l1_flag=l1, margin=1.0)
    P = {"ent_embeddings": z["ent_embeddings"], "rel_embeddings": z["rel_embeddings"]}
    train, valid, test = z["train"], z["valid"], z["test"]
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), test, R)
    cfg = hip_util.make_config(E, R, hp, train[:1], valid[:4], test[:4], optimizer="sgd", lr=0.01, batch_size=128)
    cfg.knowledge_graph.cache.update(triplets_train=train, triplets_valid=valid, triplets_test=test, hr_t=hr_t, tr_h=tr_h)
    m = hip_util.model_from_params("transe", P, hp, E, R, train=train)
    _compare("pretrained_fb15k_" + tag, m, cfg, test, z["ranks_" + tag], z["ranks64_" + tag], z["true_scores_" + tag])


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRAINED)
def test_hip_ranks_on_trained_fullsize_tables(name):
    import hip_util
    z = _fixture("ref_trained_ranks_%s.npz" % name)
    tables, m, spec, (train, valid, test), info = hip_util.train_fullsize(name)
    assert gu.tables_sha256(tables) == str(z["digest"]), (
        "the trained tables of %s are not the ones the fixture's reference ranks were computed on: the training arithmetic of the "
        "default step path (%s) changed -- regenerate (tools/make_trained_tables.py, oracle/make_golden_trained.py)" % (name, info["path"]))
    queries = z["queries"]
    hp = dict(spec["hp"])
    hp.setdefault("margin", 1.0)
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), queries, spec["R"])
    cfg = hip_util.make_config(spec["E"], spec["R"], hp, train[:1], valid[:4], queries[:4], optimizer="sgd", lr=0.01, batch_size=128)
    cfg.knowledge_graph.cache.update(triplets_train=train, triplets_valid=valid, triplets_test=test, hr_t=hr_t, tr_h=tr_h)
    m.eval()
    _compare("trained_" + name, m, cfg, queries, z["ranks"], z["ranks64"], z["true_scores"])
