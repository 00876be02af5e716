#!/usr/bin/env python
"""BASELINE.json configs C1-C4 at FULL table size through the LIVE reference (container-only; TEST INFRASTRUCTURE).

oracle/make_golden.py pins every model at toy sizes; this script runs the reference itself (read-only at
/root/reference, torch CPU fp32) at the sizes BASELINE.json quotes -- FB15k TransE d=100 (L1, L2), WN18RR ComplEx
d=200, FB15k-237 RotatE d=1000 neg 16, YAGO3-10 RESCAL k=200 -- on tables built from a numpy seed
(tests/golden_util.fullsize_inputs) and freezes ONLY the outputs into tests/golden/ref_full_<case>.npz:

  * model.forward energies of n_scores random triples                       (models/pairwise.py, pointwise.py)
  * one Trainer.train_step_* + loss.backward(): loss, dense-gradient digests  (utils/trainer.py:147-180,298)
  * Evaluator.test ranks / filtered ranks of n_rank test triples              (utils/evaluator.py:309-334)
    (none for RESCAL at YAGO3-10 size: the reference's sweep would gather E*k*k floats = 19.7 GB)

  * (`step` mode, files ref_full_step_<case>.npz) ONE optimiser step on the batch the bench's default path would draw first
    (tests/golden_util.default_step_batch: the generator's permutation rule + the Philox sampler restated on the host): loss,
    digests of every updated table and of the optimiser state (utils/trainer.py:147-180,296-299; torch.optim defaults :112-131)

  * (`ranks` mode, files ref_full_ranks_<case>.npz; round 5) a WIDE rank sample: Evaluator.test over the first 512 test triples
    of C1 (L1, L2), C2, C3 -- and, for C4 at its full E = 123 182, 16 triples whose sweeps are STITCHED from the reference's own
    Rescal.forward over 4 096-candidate chunks (each chunk's forward starts from the same pre-sweep tables, so all chunks see the
    one renormalisation a single E-candidate forward would have applied, models/pairwise.py:843-865), ranked by the reference's
    own topk + MetricCalculator scan (utils/evaluator.py:70-123,249-273).  Next to the reference's fp32 ranks the file keeps the
    FLOAT64 ranks of the same queries (numpy restatement in double): the arbiter for every query on which two fp32
    implementations disagree.

Usage: python oracle/make_golden_fullsize.py [case ...]            (scores / gradients / ranks)
       python oracle/make_golden_fullsize.py step [case ...]       (the default-path step fixtures)
       python oracle/make_golden_fullsize.py ranks [case ...]      (the wide rank samples)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
from pykg2vec.utils.trainer import Trainer  # noqa: E402
from pykg2vec.utils.evaluator import Evaluator  # noqa: E402
from pykg2vec.data.kgcontroller import Triple  # noqa: E402
import golden_util as gu  # noqa: E402
import kge_oracle as ko  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CLASS = {"transe": "pairwise.TransE", "complex": "pointwise.Complex", "rotate": "pairwise.RotatE",
         "rescal": "pairwise.Rescal"}


class _KG:
    def __init__(self, cache):
        self.cache = cache
        self.dataset_name = "synthetic"

    def read_cache_data(self, key):
        return self.cache[key]


def run(name):
    spec, P, train, valid, test, ids, batch = gu.fullsize_inputs(name)
    E, R, hp, model_name = spec["E"], spec["R"], spec["hp"], spec["model"]
    n_rank = spec["n_rank"]
    queries = test[:n_rank]
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), queries, R) if n_rank else ({}, {})
    mk = lambda arr: [Triple(int(a), int(b), int(c)) for a, b, c in arr]
    cfg = types.SimpleNamespace(
        tot_entity=E, tot_relation=R, device="cpu", optimizer="sgd", learning_rate=0.01, neg_rate=hp.get("neg_rate", 1),
        alpha=hp.get("alpha", 0.1), margin=hp.get("margin", 1.0), batch_size=spec["step_B"], tot_train_triples=len(train),
        epochs=1000, test_num=n_rank, debug=False, load_from_data=None, hits=[1, 3, 5, 10], patience=3,
        dataset_name="synthetic", sampling="uniform",
        knowledge_graph=_KG({"triplets_train": [], "triplets_valid": mk(valid[:4]), "triplets_test": mk(queries),
                             "hr_t": hr_t, "tr_h": tr_h}))
    for k, v in hp.items():
        setattr(cfg, k, v)
    cfg.summary = lambda: None
    mod, cls = CLASS[model_name].split(".")
    model_def = getattr(__import__("pykg2vec.models." + mod, fromlist=[cls]), cls)
    model = model_def(**cfg.__dict__)
    init = {k + ".weight": torch.from_numpy(v.copy()) for k, v in P.items()}
    model.load_state_dict(init)
    rec = {"name": name}

    # ---- forward energies
    model.eval()
    with torch.no_grad():
        rec["scores"] = model(*[torch.LongTensor(ids[:, i].copy()) for i in range(3)]).numpy()
    model.load_state_dict(init)  # RESCAL renormalises its tables inside forward (pairwise.py:843-844)

    # ---- one training step: loss + dense-gradient digests
    trainer = Trainer(model, cfg)
    tens = [torch.LongTensor(a) for a in batch]
    model.train()
    loss = trainer.train_step_pointwise(*tens) if model_name in gu.POINTWISE else trainer.train_step_pairwise(*tens)
    loss.backward()
    rec["loss"] = np.float32(loss.item())
    touched = np.unique(np.concatenate([batch[0], batch[2]]))[:64]
    for k, p in model.named_parameters():
        g = p.grad.numpy()
        rows = touched if g.shape[0] == E else np.unique(batch[1])[:64]
        s, a, full = gu.grad_digest(g, rows)
        rec["grad.%s.rowsum" % k], rec["grad.%s.rowabs" % k] = s, a
        rec["grad.%s.rows" % k], rec["grad.%s.full" % k] = rows, full[:, :gu.DIGEST_COLS]
    if model_name == "rescal":  # the in-place renormalisation is part of the step's observable result
        for k, v in model.state_dict().items():
            rec["after_fwd.%s.rows" % k] = v.numpy()[:64, :gu.DIGEST_COLS].copy()

    # ---- Evaluator.test on the first n_rank test triples
    if n_rank:
        model.load_state_dict(init)
        ev = Evaluator(model, cfg)
        model.eval()
        with torch.no_grad():
            ev.test(ev.test_data, n_rank, epoch=0)
        mc = ev.metric_calculator
        rec["ranks"] = np.stack([np.asarray(x, np.int64) for x in (mc.rank_head, mc.rank_tail, mc.f_rank_head, mc.f_rank_tail)])
        # the reference's own energy of the true candidate and how many candidates sit inside the fp32 band around it
        with torch.no_grad():
            ents = torch.arange(E)
            st = []
            for h, r, t in queries:
                sh = model(ents, torch.full((E,), int(r)), torch.full((E,), int(t))).numpy()
                stl = model(torch.full((E,), int(h)), torch.full((E,), int(r)), ents).numpy()
                st.append((sh[int(h)], stl[int(t)]))
        rec["true_scores"] = np.asarray(st, np.float32)   # [n_rank, 2] = (head sweep, tail sweep)
    np.savez_compressed(os.path.join(OUT, "ref_full_%s.npz" % name), **rec)
    print("wrote", name, "loss=%.6f" % rec["loss"], "ranks" if n_rank else "", rec.get("ranks", np.zeros(0))[:, :4] if n_rank else "")


def run_step(name):
    """One reference step (zero_grad, train_step_*, backward, optimizer.step) on the default path's first batch."""
    spec, step, P, train, pos, (nh, nr, nt) = gu.default_step_batch(name)
    E, R, hp, model_name = spec["E"], spec["R"], spec["hp"], spec["model"]
    neg_rate = hp.get("neg_rate", 1)
    cfg = types.SimpleNamespace(
        tot_entity=E, tot_relation=R, device="cpu", optimizer=step["optimizer"], learning_rate=step["lr"], neg_rate=neg_rate,
        alpha=hp.get("alpha", 0.1), margin=hp.get("margin", 1.0), batch_size=step["B"], tot_train_triples=len(train),
        epochs=1000, test_num=0, debug=False, load_from_data=None, hits=[1, 3, 5, 10], patience=3,
        dataset_name="synthetic", sampling="uniform",
        knowledge_graph=_KG({"triplets_train": [], "triplets_valid": [], "triplets_test": [], "hr_t": {}, "tr_h": {}}))
    for k, v in hp.items():
        setattr(cfg, k, v)
    cfg.summary = lambda: None
    mod, cls = CLASS[model_name].split(".")
    model_def = getattr(__import__("pykg2vec.models." + mod, fromlist=[cls]), cls)
    torch.manual_seed(0)
    model = model_def(**cfg.__dict__)
    model.load_state_dict({k + ".weight": torch.from_numpy(v.copy()) for k, v in P.items()})
    trainer = Trainer(model, cfg)
    trainer.build_model()                                       # utils/trainer.py:103-144: model.to(device), torch.optim.<kind>(lr)
    if model_name in gu.POINTWISE:
        batch = ko.pointwise_layout(pos, nh, nr, nt, neg_rate)
    else:
        batch = (pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt)
    tens = [torch.LongTensor(np.ascontiguousarray(a)) for a in batch]
    model.train()
    trainer.optimizer.zero_grad()                               # utils/trainer.py:296-299
    loss = trainer.train_step_pointwise(*tens) if model_name in gu.POINTWISE else trainer.train_step_pairwise(*tens)
    loss.backward()
    trainer.optimizer.step()
    rec = {"name": name, "loss": np.float32(loss.item()), "B": np.int64(step["B"]),
           "batch_checksum": np.int64(sum(int(np.asarray(a, np.int64).sum()) * (i + 1) for i, a in enumerate(batch)))}
    touched_e = np.unique(np.concatenate([pos[:, 0], pos[:, 2], nh, nt]))
    untouched_e = np.setdiff1d(np.arange(E), touched_e)[:32]
    touched_r = np.unique(pos[:, 1])
    untouched_r = np.setdiff1d(np.arange(R), touched_r)[:32]
    state = trainer.optimizer.state
    for k, p in model.named_parameters():
        is_ent = p.shape[0] == E
        rows = np.concatenate([touched_e[:64], untouched_e]) if is_ent else np.concatenate([touched_r[:64], untouched_r])
        rec["rows.%s" % k] = rows
        for label, tensor in (("post", p.detach()),) + tuple((kind, state[p][key]) for kind, key in
                                                              (("state1", "exp_avg" if step["optimizer"] == "adam" else
                                                                "sum" if step["optimizer"] == "adagrad" else "square_avg"),
                                                               ("state2", "exp_avg_sq")) if key in state[p]):
            s_, a_, full = gu.table_digest(tensor.numpy(), rows)
            rec["%s.%s.rowsum" % (label, k)], rec["%s.%s.rowabs" % (label, k)], rec["%s.%s.rows" % (label, k)] = s_, a_, full
    np.savez_compressed(os.path.join(OUT, "ref_full_step_%s.npz" % name), **rec)
    print("wrote step", name, "B=%d %s loss=%.6f" % (step["B"], step["optimizer"], rec["loss"]))


WIDE_RANKS = {"c1_transe_l1": 512, "c1_transe_l2": 512, "c2_complex": 512, "c3_rotate": 512, "c4_rescal": 16}
RESCAL_CHUNK = 4096


def _ranks64(spec, P, queries, hr_t, tr_h):
    """Float64 ranks of the same queries: rank = #{e : s_e < s_true} in double (ties in double do not occur on these tables)."""
    model_name, hp = spec["model"], dict(spec["hp"])
    hp.setdefault("margin", 1.0)
    out = np.zeros((4, len(queries)), np.int64)
    if model_name == "rescal":      # h' M_r t over all entities without materialising E x k x k
        k = hp["hidden_size"]
        P64 = {n: v.astype(np.float64) for n, v in P.items()}
        ent = P64["ent_embeddings"] / np.linalg.norm(P64["ent_embeddings"], axis=1, keepdims=True)
        rel = P64["rel_matrices"] / np.linalg.norm(P64["rel_matrices"], axis=1, keepdims=True)
        for i, (h, r, t) in enumerate(queries):
            M = rel[r].reshape(k, k)
            sh = -(ent @ (M @ ent[t]))          # (e, r, t) for every e
            st = -(ent @ (ent[h] @ M))          # (h, r, e)
            out[0, i], out[2, i] = ko.rank_from_scores(sh, int(h), tr_h[(int(t), int(r))])
            out[1, i], out[3, i] = ko.rank_from_scores(st, int(t), hr_t[(int(h), int(r))])
        return out
    for i, (h, r, t) in enumerate(queries):
        h, r, t = int(h), int(r), int(t)
        sh = ko.sweep_scores(model_name, P, h, r, t, "head", dtype=np.float64, **hp)
        st = ko.sweep_scores(model_name, P, h, r, t, "tail", dtype=np.float64, **hp)
        out[0, i], out[2, i] = ko.rank_from_scores(sh, h, tr_h[(t, r)])
        out[1, i], out[3, i] = ko.rank_from_scores(st, t, hr_t[(h, r)])
    return out


def _stitched_rescal(model, E, queries, hr_t, tr_h):
    """Evaluator.test for a model whose single E-candidate forward does not fit memory: the same sweeps, chunk by chunk, through the
    reference's own forward; the ordering and the rank scan are the reference's (torch.topk(k=E), MetricCalculator)."""
    from pykg2vec.utils.evaluator import MetricCalculator
    cfg = types.SimpleNamespace(knowledge_graph=_KG({"hr_t": hr_t, "tr_h": tr_h}), hits=[1, 3, 5, 10])
    mc = MetricCalculator(cfg)
    true_scores = []

    def sweep(fixed_a, fixed_b, tail):
        before = {k: v.clone() for k, v in model.state_dict().items()}
        parts, after = [], None
        for lo in range(0, E, RESCAL_CHUNK):
            if after is not None:                       # every chunk starts from the pre-sweep tables
                model.ent_embeddings.weight.data = before["ent_embeddings.weight"].clone()
                model.rel_matrices.weight.data = before["rel_matrices.weight"].clone()
            ents = torch.arange(lo, min(E, lo + RESCAL_CHUNK))
            a = torch.LongTensor([fixed_a]).repeat([len(ents)])
            b = torch.LongTensor([fixed_b]).repeat([len(ents)])
            parts.append(model.forward(a, b, ents) if tail else model.forward(ents, a, b))
            after = True
        preds = torch.cat(parts)
        return preds, torch.topk(preds, k=E)[1]

    with torch.no_grad():
        for i, (h, r, t) in enumerate(queries):
            h, r, t = int(h), int(r), int(t)
            ph, hrank = sweep(r, t, tail=False)         # utils/evaluator.py:321-322: head sweep first
            pt, trank = sweep(h, r, tail=True)
            true_scores.append((ph[h].item(), pt[t].item()))
            mc.append_result([trank.numpy(), hrank.numpy(), h, r, t, 0])
            print("  stitched query %d/%d" % (i + 1, len(queries)), flush=True)
    return mc, np.asarray(true_scores, np.float32)


def run_ranks(name):
    spec, P, train, valid, test, ids, batch = gu.fullsize_inputs(name)
    E, R, hp, model_name = spec["E"], spec["R"], spec["hp"], spec["model"]
    n = WIDE_RANKS[name]
    queries = test[:n]
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), queries, R)
    mk = lambda arr: [Triple(int(a), int(b), int(c)) for a, b, c in arr]
    cfg = types.SimpleNamespace(
        tot_entity=E, tot_relation=R, device="cpu", optimizer="sgd", learning_rate=0.01, neg_rate=hp.get("neg_rate", 1),
        alpha=hp.get("alpha", 0.1), margin=hp.get("margin", 1.0), batch_size=spec["step_B"], tot_train_triples=len(train),
        epochs=1000, test_num=n, debug=False, load_from_data=None, hits=[1, 3, 5, 10], patience=3,
        dataset_name="synthetic", sampling="uniform",
        knowledge_graph=_KG({"triplets_train": [], "triplets_valid": mk(valid[:4]), "triplets_test": mk(queries),
                             "hr_t": hr_t, "tr_h": tr_h}))
    for k, v in hp.items():
        setattr(cfg, k, v)
    cfg.summary = lambda: None
    mod, cls = CLASS[model_name].split(".")
    model_def = getattr(__import__("pykg2vec.models." + mod, fromlist=[cls]), cls)
    model = model_def(**cfg.__dict__)
    model.load_state_dict({k + ".weight": torch.from_numpy(v.copy()) for k, v in P.items()})
    model.eval()
    rec = {"name": name, "n": np.int64(n)}
    if model_name == "rescal":
        mc, rec["true_scores"] = _stitched_rescal(model, E, queries, hr_t, tr_h)
        rec["chunk"] = np.int64(RESCAL_CHUNK)
    else:
        import contextlib
        import io
        ev = Evaluator(model, cfg)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ev.test(ev.test_data, n, epoch=0)
        mc = ev.metric_calculator
        with torch.no_grad():
            ents = torch.arange(E)
            st = []
            for h, r, t in queries:
                sh = model(ents, torch.full((E,), int(r)), torch.full((E,), int(t)))
                stl = model(torch.full((E,), int(h)), torch.full((E,), int(r)), ents)
                st.append((sh[int(h)].item(), stl[int(t)].item()))
        rec["true_scores"] = np.asarray(st, np.float32)
    rec["ranks"] = np.stack([np.asarray(x, np.int64) for x in (mc.rank_head, mc.rank_tail, mc.f_rank_head, mc.f_rank_tail)])
    print("reference ranks done:", name, flush=True)
    rec["ranks64"] = _ranks64(spec, P, queries, hr_t, tr_h)
    np.savez_compressed(os.path.join(OUT, "ref_full_ranks_%s.npz" % name), **rec)
    diff = int((rec["ranks"] != rec["ranks64"]).any(0).sum())
    print("wrote ranks", name, "n=%d" % n, "queries where the reference's fp32 ranks differ from float64: %d" % diff, flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "ranks":
        for name in (args[1:] or list(WIDE_RANKS)):
            run_ranks(name)
    elif args and args[0] == "step":
        for name in (args[1:] or list(gu.DEFAULT_STEP)):
            run_step(name)
    else:
        for name in (args or list(gu.FULLSIZE)):
            run(name)
