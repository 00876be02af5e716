"""BASELINE.json configs C1-C4 at FULL table size against outputs of the LIVE reference
(tests/golden/ref_full_*.npz, frozen by oracle/make_golden_fullsize.py; inputs re-created from the seed by
golden_util.fullsize_inputs).  CPU half: the numpy oracle is held to the reference at these sizes.  GPU half: the HIP
path (forward, one fused training step, filtered ranks) is held to the same files; every rank that differs from the
reference's must be explained by candidates inside the fp32 tolerance band, and the observed agreement is written to
gpurun_out/rank_agreement_fullsize.json (copied to profiles/ by the builder)."""
import json
import os

import numpy as np
import pytest

import kge_oracle as ko
import golden_util as gu
from golden_util import FULLSIZE, GOLDEN, close

CASES = list(FULLSIZE)
_INPUTS = {}


def _inputs(name):
    if name not in _INPUTS:
        _INPUTS.clear()  # one case resident at a time (C4 tables are ~130 MB)
        _INPUTS[name] = gu.fullsize_inputs(name)
    return _INPUTS[name]


def _golden(name):
    return np.load(os.path.join(GOLDEN, "ref_full_%s.npz" % name))


def _hp(spec):
    hp = dict(spec["hp"])
    hp.setdefault("margin", 1.0)
    return hp


def _check_grad_digest(z, key, g, scale_floor=1e-3):
    rows = z["grad.%s.rows" % key]
    s, a, full = gu.grad_digest(g, rows)
    ref_s, ref_a, ref_full = z["grad.%s.rowsum" % key], z["grad.%s.rowabs" % key], z["grad.%s.full" % key]
    scale = max(scale_floor, float(np.abs(ref_full).max()))
    assert np.allclose(full[:, :gu.DIGEST_COLS], ref_full, atol=2e-5 * max(1.0, scale), rtol=1e-3), \
        (key, np.abs(full[:, :gu.DIGEST_COLS] - ref_full).max())
    # per-row digests over ALL rows: untouched rows are exactly zero in both, touched rows agree to fp32 noise
    assert np.array_equal(ref_a == 0, a == 0), key
    d = g.shape[1]
    assert np.allclose(a, ref_a, atol=1e-5 * d * max(1.0, scale), rtol=1e-3), (key, np.abs(a - ref_a).max())
    assert np.allclose(s, ref_s, atol=1e-5 * d * max(1.0, scale), rtol=1e-3), (key, np.abs(s - ref_s).max())


# ------------------------------------------------------------------ CPU: oracle vs the live reference at full size
@pytest.mark.parametrize("name", CASES)
def test_oracle_forward_and_step_match_reference_at_full_size(name):
    spec, P, train, valid, test, ids, batch = _inputs(name)
    z = _golden(name)
    model, hp = spec["model"], _hp(spec)
    Pn = ko.rescal_normalize_tables(P) if model == "rescal" else P
    got = ko.score(model, Pn, ids[:, 0], ids[:, 1], ids[:, 2], **hp)
    assert close(got, z["scores"]), np.abs(got - z["scores"]).max()
    loss, G, _, Pafter = ko.train_step_grads(model, {k: v.copy() for k, v in P.items()}, batch, **hp)
    assert np.isclose(loss, z["loss"], rtol=2e-5, atol=2e-5), (loss, z["loss"])
    for k, g in G.items():
        _check_grad_digest(z, k + ".weight", g)
    if model == "rescal":
        for k, v in Pafter.items():
            assert close(v[:64, :gu.DIGEST_COLS], z["after_fwd.%s.weight.rows" % k])


@pytest.mark.parametrize("name", [n for n in CASES if FULLSIZE[n]["n_rank"]])
def test_oracle_ranks_match_reference_at_full_size(name):
    spec, P, train, valid, test, ids, batch = _inputs(name)
    z = _golden(name)
    n = spec["n_rank"] if spec["model"] != "rotate" else 2   # d=1000 numpy sweeps: keep the CPU suite short
    q = test[:n]
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), test[:spec["n_rank"]], spec["R"])
    ref = z["ranks"][:, :n]
    hp = _hp(spec)
    if spec["model"] == "rescal":   # the reference's forward renormalises both tables in place before scoring (pairwise.py:843-844)
        P = ko.rescal_normalize_tables(P)
    for i, (h, r, t) in enumerate(q):
        h, r, t = int(h), int(r), int(t)
        sh = ko.sweep_scores(spec["model"], P, h, r, t, "head", **hp)
        st = ko.sweep_scores(spec["model"], P, h, r, t, "tail", **hp)
        assert close(sh[h], z["true_scores"][i, 0]) and close(st[t], z["true_scores"][i, 1])
        for s, true, known, raw, filt in ((sh, h, tr_h[(t, r)], ref[0, i], ref[2, i]), (st, t, hr_t[(h, r)], ref[1, i], ref[3, i])):
            rk, frk = ko.rank_from_scores(s, true, known)
            ok_r, _ = gu.rank_band_ok(s, true, rk, raw)
            ok_f, _ = gu.rank_band_ok(s, true, frk, filt)
            assert ok_r and ok_f, (name, i, rk, raw, frk, filt)


# ------------------------------------------------------------------ GPU: the HIP path vs the live reference at full size
def _gpu_setup(name, batch_size=None):
    import hip_util
    from pykg2vec_amd.trainer import Trainer
    spec, P, train, valid, test, ids, batch = _inputs(name)
    hp = _hp(spec)
    n_rank = spec["n_rank"]
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), test[:n_rank], spec["R"]) if n_rank else ({}, {})
    cfg = hip_util.make_config(spec["E"], spec["R"], hp, train[:1], valid[:4], test[:max(n_rank, 1)], optimizer="sgd", lr=0.01,
                               batch_size=batch_size or spec["step_B"])
    # (the three full splits: the Evaluator builds its filter lists from them on the device; hr_t / tr_h hold the same sets for the queries)
    cfg.knowledge_graph.cache.update(triplets_train=train, triplets_valid=valid, triplets_test=test, hr_t=hr_t, tr_h=tr_h)
    cfg.tot_train_triples = len(train)
    m = hip_util.model_from_params(spec["model"], P, spec["hp"], spec["E"], spec["R"], train=train)
    return hip_util, spec, hp, cfg, m, Trainer


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_forward_and_fused_step_match_reference_at_full_size(name):
    import torch
    hip, spec, hp, cfg, m, Trainer = _gpu_setup(name)
    _, P, train, valid, test, ids, batch = _inputs(name)
    z = _golden(name)
    with torch.no_grad():
        got = m(hip.dev(ids[:, 0]), hip.dev(ids[:, 1]), hip.dev(ids[:, 2])).cpu().numpy()
    if spec["model"] == "rescal":   # forward renormalised the tables in place (pairwise.py:843-844): restore
        m = hip.model_from_params(spec["model"], P, spec["hp"], spec["E"], spec["R"], train=train)
    assert close(got, z["scores"]), np.abs(got - z["scores"]).max()
    tr = Trainer(m, cfg, use_graph=False)
    tr.build_model()
    b = [hip.dev(x) for x in batch]
    loss = tr.train_step_pointwise(*b) if spec["model"] in gu.POINTWISE else tr.train_step_pairwise(*b)
    assert np.isclose(loss.item(), z["loss"], rtol=3e-5, atol=3e-5), (loss.item(), z["loss"])
    for (pname, _), g in zip(hip.table_parameters(m), tr.flat.grad_views):
        _check_grad_digest(z, pname, g.cpu().numpy())
    if spec["model"] == "rescal":
        for pname, p in hip.table_parameters(m):
            assert close(p.detach().cpu().numpy()[:64, :gu.DIGEST_COLS], z["after_fwd.%s.rows" % pname])


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in CASES if FULLSIZE[n]["n_rank"]])
def test_hip_ranks_match_reference_at_full_size_inside_the_score_band(name):
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    hip, spec, hp, cfg, m, _ = _gpu_setup(name)
    _, P, train, valid, test, ids, batch = _inputs(name)
    z = _golden(name)
    n = spec["n_rank"]
    q = test[:n]
    ranks = Evaluator(m, cfg).rank_all(q, n).cpu().numpy()
    ref = z["ranks"]
    scores = K.eval_sweep_scores(m.make_desc(), hip.dev(q)).cpu().numpy()   # [2n, E]: tail sweep, head sweep per triple
    report = {"case": name, "queries": 2 * n, "raw_equal": 0, "filtered_equal": 0, "max_abs_rank_diff": 0, "flips": []}
    for i, (h, r, t) in enumerate(q):
        for side, row, true, raw_g, filt_g, raw_r, filt_r in (
                ("tail", scores[2 * i], int(t), ranks[1, i], ranks[3, i], ref[1, i], ref[3, i]),
                ("head", scores[2 * i + 1], int(h), ranks[0, i], ranks[2, i], ref[0, i], ref[2, i])):
            assert close(row[true], z["true_scores"][i, 0 if side == "head" else 1], atol=2e-5, rtol=2e-5)
            ok_r, near = gu.rank_band_ok(row, true, raw_g, raw_r)
            ok_f, _ = gu.rank_band_ok(row, true, filt_g, filt_r)
            assert ok_r and ok_f, (name, i, side, int(raw_g), int(raw_r), int(filt_g), int(filt_r), near)
            report["raw_equal"] += int(raw_g == raw_r)
            report["filtered_equal"] += int(filt_g == filt_r)
            report["max_abs_rank_diff"] = max(report["max_abs_rank_diff"], abs(int(raw_g) - int(raw_r)))
            if raw_g != raw_r or filt_g != filt_r:
                report["flips"].append({"triple": i, "side": side, "gpu": [int(raw_g), int(filt_g)],
                                        "reference": [int(raw_r), int(filt_r)], "candidates_inside_band": near})
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "rank_agreement_fullsize.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[name] = report
    json.dump(doc, open(path, "w"), indent=1)


# ------------------------------------------------------------------ ONE optimiser step of every BASELINE config through the path the
# bench times, against the live reference (tests/golden/ref_full_step_*.npz, oracle/make_golden_fullsize.py step)
STEP_CASES = list(gu.DEFAULT_STEP)
GRAD_FLOOR_REL = 1e-4  # the first Adam / Adagrad step is p -= lr * g / (|g| + eps) = lr * sign(g): an element whose gradient is fp32
                       # noise (|g| below this fraction of the table's largest gradient element; implementations agree on g to ~1e-6
                       # of that) may step the other way.  Such elements are held to |delta p| <= 2 lr and counted in the report.
UPDATED_ROW_ATOL = {}                      # (round 4: {"c4_rescal": 5e-7} -- RESCAL's entity gradients were float atomics; round 5 stages them
                                           #  per (pair, side) and sums them in slot order: C4's default path is deterministic like the others)
UPDATED_ROW_ATOL_DET = 2e-7                # every default path sums in a fixed order: 2x the largest deviation observed (profiles/r05_step_agreement_fullsize.json: 9e-8)


def _step_golden(name):
    return np.load(os.path.join(GOLDEN, "ref_full_step_%s.npz" % name))


def _grad_from_state1(optimizer, s1):
    return {"adam": lambda m: m / 0.1, "adagrad": np.sqrt, "rms": lambda s: np.sqrt(s / 0.01)}[optimizer](np.asarray(s1, np.float64))


def check_step_against_reference(name, loss, tables, state1, state2, report, loss_rtol=3e-5, state_rtol=1e-4, row_atol=None,
                                 state_floor=5e-5, state2_rtol=None):
    """tables / state1 / state2: {state_dict key: array}.  Untouched rows must equal the reference's bit for bit (dense optimisers
    with zero gradient and zero state leave a row unchanged); gradient-carrying quantities (optimiser state) agree to fp32 noise over
    ALL rows (float64 row digests); the listed rows of the updated tables agree element-wise wherever the gradient is not noise."""
    z = _step_golden(name)
    step = gu.DEFAULT_STEP[name]
    lr, opt = step["lr"], step["optimizer"]
    assert np.isclose(loss, z["loss"], rtol=loss_rtol, atol=loss_rtol), (loss, float(z["loss"]))
    rep = {"loss": float(loss), "loss_reference": float(z["loss"]), "tables": {}}
    for key, w in tables.items():
        rows = z["rows.%s" % key]
        ref_rows = z["post.%s.rows" % key]
        got_rows = np.asarray(w)[rows][:, :gu.DIGEST_COLS]
        ref_s1 = z["state1.%s.rows" % key]
        got_s1 = np.asarray(state1[key])[rows][:, :gu.DIGEST_COLS]
        scale1 = max(float(np.abs(ref_s1).max()), 1e-12)
        d1 = np.abs(got_s1 - ref_s1)
        # (Adagrad / RMSprop keep SQUARED gradients: twice the relative error; relation rows sum thousands of cancelling terms)
        assert np.all(d1 <= state_rtol * np.abs(ref_s1) + state_floor * scale1), (key, "state1 rows", float(d1.max()), scale1)
        g = np.abs(_grad_from_state1(opt, ref_s1))
        solid = g >= GRAD_FLOOR_REL * max(float(g.max()), 1e-30)
        dp = np.abs(got_rows - ref_rows)
        untouched = ~np.any(ref_s1 != 0, axis=1)
        # rows nobody touched: bit-identical -- except RESCAL, whose forward renormalises every row (same value to rounding)
        if name == "c4_rescal":
            assert np.all(dp[untouched] <= 2e-6), (key, "untouched rows", float(dp[untouched].max()))
        else:
            assert np.array_equal(got_rows[untouched], ref_rows[untouched]), (key, "untouched rows moved")
        solid = solid | untouched[:, None]
        # the first step moves an element by lr * g / (|g| + eps): an ABSOLUTE gradient error e (cancellation noise, ~1e-6 of the
        # table's largest gradient element) shifts it by at most lr * e / |g|, which matters only for the smallest elements
        gmax = max(float(g.max()), 1e-30)
        tol_solid = (row_atol if row_atol is not None else UPDATED_ROW_ATOL.get(name, UPDATED_ROW_ATOL_DET)) + lr * 1e-6 * gmax / np.maximum(g, 1e-30)
        tol_solid = np.where(untouched[:, None], 2e-6 if name == "c4_rescal" else 0.0, tol_solid)
        assert np.all(dp[solid] <= tol_solid[solid]), (key, "post rows", float((dp - tol_solid)[solid].max()))
        assert np.all(dp[~solid] <= 2.0 * lr * 1.001 + 1e-6), (key, "noise-gradient elements", float(dp[~solid].max()))
        # digests over ALL rows of the gradient-carrying state
        got_sum, got_abs, _ = gu.table_digest(state1[key], rows[:1])
        ref_abs = z["state1.%s.rowabs" % key]
        assert np.array_equal(ref_abs == 0, got_abs == 0), (key, "set of rows with optimiser state differs")
        assert np.allclose(got_abs, ref_abs, rtol=state_rtol, atol=2e-6 * scale1 * w.shape[1]), (key, float(np.abs(got_abs - ref_abs).max()))
        entry = {"max_abs_diff_updated_rows": float(dp[solid].max()) if solid.any() else 0.0,
                 "max_abs_diff_state1_rows": float(d1.max()), "state1_scale": scale1,
                 "noise_gradient_elements": int((~solid).sum()), "elements_compared": int(solid.size),
                 "max_rel_diff_state1_rowabs_all_rows": float(np.max(np.abs(got_abs - ref_abs) / np.maximum(ref_abs, 1e-30) * (ref_abs > 0)))}
        if state2 is not None and ("state2.%s.rows" % key) in z.files:
            ref_s2 = z["state2.%s.rows" % key]
            got_s2 = np.asarray(state2[key])[rows][:, :gu.DIGEST_COLS]
            scale2 = max(float(np.abs(ref_s2).max()), 1e-30)
            d2 = np.abs(got_s2 - ref_s2)
            assert np.all(d2 <= 2 * (state2_rtol or state_rtol) * np.abs(ref_s2) + 4e-6 * scale2), (key, "state2 rows", float(d2.max()), scale2)
            entry["max_abs_diff_state2_rows"] = float(d2.max())
        rep["tables"][key] = entry
    report[name] = rep
    return rep


@pytest.mark.parametrize("name", STEP_CASES)
def test_oracle_default_step_matches_reference_at_full_size(name):
    """The numpy oracle on the restated first batch: loss, updated tables and optimiser state of the live reference."""
    spec, step, P, train, pos, (nh, nr, nt) = gu.default_step_batch(name)
    z = _step_golden(name)
    model, hp = spec["model"], _hp(spec)
    neg_rate = hp.get("neg_rate", 1)
    batch = ko.pointwise_layout(pos, nh, nr, nt, neg_rate) if model in gu.POINTWISE else (pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt)
    assert int(sum(int(np.asarray(a, np.int64).sum()) * (i + 1) for i, a in enumerate(batch))) == int(z["batch_checksum"])
    loss, G, _, Pn = ko.train_step_grads(model, {k: v.copy() for k, v in P.items()}, batch, **hp)
    Pn = {k: np.array(v, copy=True) for k, v in Pn.items()}
    st = ko.optimizer_init(step["optimizer"], Pn)
    ko.optimizer_step(step["optimizer"], Pn, G, st, step["lr"])
    s1 = st["m"] if step["optimizer"] == "adam" else st["sq"]
    wkey = lambda d: {k + ".weight": v for k, v in d.items()}
    # (numpy's reduction order differs from both ATen's and the kernels': a looser row tolerance than the GPU paths are held to)
    check_step_against_reference(name, loss, wkey(Pn), wkey(s1), wkey(st["v"]) if step["optimizer"] == "adam" else None, {},
                                 state_rtol=2e-4, row_atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", STEP_CASES)
def test_hip_default_path_step_matches_reference_at_full_size(name):
    """Trainer.train_model_epoch over ONE batch of the bench's size: whatever step path the trainer picks by default for the config
    (owner-computes two-phase step, staged own-step, staged RotatE step, hipGraph-captured RESCAL pair step + row-owner optimiser),
    its device sampler included, against the reference's step on the same batch."""
    import torch
    import hip_util
    from pykg2vec_amd.trainer import Trainer
    spec, P, train, valid, test, _ids, _batch = _inputs(name)
    step = gu.DEFAULT_STEP[name]
    hp = _hp(spec)
    cfg = hip_util.make_config(spec["E"], spec["R"], hp, train[:1], valid[:4], test[:4], optimizer=step["optimizer"], lr=step["lr"],
                               batch_size=step["B"])
    cfg.knowledge_graph.cache.update(triplets_train=train)
    cfg.seed, cfg.tot_train_triples, cfg.sampling = gu.GENERATOR_SEED, step["B"], "uniform"
    m = hip_util.model_from_params(spec["model"], P, spec["hp"], spec["E"], spec["R"], train=train)
    tr = Trainer(m, cfg)
    tr.build_model()
    tr.generator = tr._new_generator()
    loss = tr.train_model_epoch(0)
    tr.sync_model()
    torch.cuda.synchronize()
    path = ("hipGraph" if tr._graph is not None else "staged" if getattr(tr, "_staged", None) is not None else
            "own" if getattr(tr, "_own", None) is not None else "pull" if getattr(tr, "_pull", None) is not None else "eager")
    want = {"c1_transe_l1": "pull", "c1_transe_l2": "pull", "c2_complex": "own", "c3_rotate": "staged", "c4_rescal": "hipGraph"}[name]
    assert path == want, (name, path)
    flat = tr.flat
    named = hip_util.table_parameters(m)
    off = [v.data_ptr() - flat.param.data_ptr() for v in flat.views]
    view = lambda buf, o, v: buf[o // 4:o // 4 + v.numel()].view_as(v).cpu().numpy()
    tables = {k: p.detach().cpu().numpy() for k, p in named}
    s1 = {k: view(flat.state1, o, v) for (k, _), o, v in zip(named, off, flat.views)}
    s2 = {k: view(flat.state2, o, v) for (k, _), o, v in zip(named, off, flat.views)} if flat.state2 is not None else None
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    fn = os.path.join(out, "step_agreement_fullsize.json")
    doc = json.load(open(fn)) if os.path.exists(fn) else {}
    # round 6 ratchet: ~3x the largest deviations on record (profiles/r05_step_agreement_fullsize.json,
    # r06_step_agreement_fullsize.json) -- loss 1.2e-7 relative, row digests of the optimiser state 3.2e-6 relative; the element-wise
    # floor stays at 5e-5 of the state's scale: C2's relation rows (Adagrad: SQUARED sums of ~900 cancelling terms) sit at 2.4e-5
    rep = check_step_against_reference(name, loss, tables, s1, s2, doc, loss_rtol=5e-7, state_rtol=1e-5, state_floor=5e-5, state2_rtol=1e-4)
    rep["path"] = path + (", two-phase" if path == "pull" and tr._pull.direction is not None else "")
    json.dump(doc, open(fn, "w"), indent=1)


# ------------------------------------------------------------------ WIDE rank samples (round 5): 512 test triples per config -- and 16 at
# C4's full E = 123 182, stitched from the reference's own Rescal.forward over 4 096-candidate chunks -- with the float64 ranks of the same
# queries as arbiter (tests/golden/ref_full_ranks_*.npz, oracle/make_golden_fullsize.py ranks)
WIDE = {"c1_transe_l1": 512, "c1_transe_l2": 512, "c2_complex": 512, "c3_rotate": 512, "c4_rescal": 16}


def _wide_golden(name):
    path = os.path.join(GOLDEN, "ref_full_ranks_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated" % os.path.basename(path))
    return np.load(path)


@pytest.mark.parametrize("name", list(WIDE))
def test_wide_rank_fixture_is_consistent_with_float64(name):
    """The reference's fp32 ranks and the float64 ranks of the same queries differ only by near-ties: a handful of queries, small steps."""
    z = _wide_golden(name)
    ref, r64 = z["ranks"], z["ranks64"]
    assert ref.shape == (4, WIDE[name]) and r64.shape == ref.shape
    differ = (ref != r64).any(0)
    # RotatE at d = 1000 sums 2 000 squares per energy on near-uniform random tables: 30 % of the triples have a candidate within fp32
    # noise of the true one (the reference's own fp32 ranks move by 1-2 there); the other configs a few per cent
    assert differ.mean() <= (0.4 if name == "c3_rotate" else 0.06), differ.mean()
    assert np.abs(ref - r64).max() <= 4
    assert (ref[2] <= ref[0]).all() and (ref[3] <= ref[1]).all()      # filtered <= raw


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(WIDE))
def test_hip_ranks_on_the_wide_sample_with_float64_arbitration(name):
    from pykg2vec_amd import kernels as K
    from pykg2vec_amd.evaluator import Evaluator
    import hip_util
    z = _wide_golden(name)
    n = WIDE[name]
    spec, P, train, valid, test, ids, batch = _inputs(name)
    hp = _hp(spec)
    q = test[:n]
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), q, spec["R"])
    cfg = hip_util.make_config(spec["E"], spec["R"], hp, train[:1], valid[:4], q, optimizer="sgd", lr=0.01, batch_size=spec["step_B"])
    cfg.knowledge_graph.cache.update(triplets_train=train, triplets_valid=valid, triplets_test=test, hr_t=hr_t, tr_h=tr_h)
    m = hip_util.model_from_params(spec["model"], P, spec["hp"], spec["E"], spec["R"], train=train)
    ranks = Evaluator(m, cfg).rank_all(q, n).cpu().numpy()
    ref, r64 = z["ranks"], z["ranks64"]
    differ = np.flatnonzero((ranks != ref).any(0))
    report = {"case": name, "test_triples": n, "queries": 2 * n, "triples_with_a_differing_rank": int(len(differ)),
              "hip_equals_float64": 0, "reference_equals_float64": 0, "neither": 0, "max_abs_rank_diff": 0, "flips": []}
    if len(differ):
        scores = K.eval_sweep_scores(m.make_desc(), hip_util.dev(q[differ])).cpu().numpy()   # [2 d, E]: tail sweep, head sweep per triple
    for j, i in enumerate(differ):
        h, r, t = (int(x) for x in q[i])
        for side, row, true, (a, b) in (("tail", scores[2 * j], t, (1, 3)), ("head", scores[2 * j + 1], h, (0, 2))):
            assert close(row[true], z["true_scores"][i, 0 if side == "head" else 1], atol=2e-5, rtol=2e-5)
            for which in (a, b):
                g_, r_, d_ = int(ranks[which, i]), int(ref[which, i]), int(r64[which, i])
                if g_ == r_:
                    continue
                ok, near = gu.rank_band_ok(row, true, g_, r_)
                assert ok, (name, int(i), side, which, g_, r_, near)
                report["max_abs_rank_diff"] = max(report["max_abs_rank_diff"], abs(g_ - r_))
                verdict = "hip" if g_ == d_ else ("reference" if r_ == d_ else "neither")
                report["hip_equals_float64" if verdict == "hip" else "reference_equals_float64" if verdict == "reference" else "neither"] += 1
                report["flips"].append({"triple": int(i), "side": side, "filtered": which >= 2, "hip": g_, "reference": r_, "float64": d_,
                                        "float64_sides_with": verdict, "candidates_inside_band": near})
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "rank_agreement_fullsize_wide.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[name] = report
    json.dump(doc, open(path, "w"), indent=1)
    # near-ties are rare on these tables except RotatE d = 1000 (see the fixture's own fp32-vs-float64 count); a systematic deviation
    # would flip many more.  Round 6 ratchet: ~2x the observed share of triples with a differing rank (profiles/r06_rank_agreement_wide.json:
    # C1 1.6 % / 0.8 %, C2 0.2 % -- 1.0 % before the chunked k chain --, C3 31 %, C4 0 of 16)
    limit = {"c1_transe_l1": 0.03, "c1_transe_l2": 0.03, "c2_complex": 0.02, "c3_rotate": 0.40, "c4_rescal": 1.0 / 16}[name]
    assert len(differ) <= limit * n, (len(differ), n)
    if name == "c2_complex":   # no longer one-sided against the HIP path by a margin (round 5: float64 sided with the reference 10 : 0)
        assert report["reference_equals_float64"] <= 4, report
