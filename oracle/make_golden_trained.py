#!/usr/bin/env python
"""Rank fixtures on TRAINED tables through the LIVE reference (container-only; TEST INFRASTRUCTURE).

Every full-size rank fixture of rounds 4-5 (make_golden_fullsize.py) sits on freshly initialised tables, where candidates
within fp32 noise of the true one are densest.  This script freezes the reference's Evaluator.test ranks
(utils/evaluator.py:70-123,249-273,309-334) -- next to the float64 ranks of the same queries, the arbiter wherever two fp32
implementations disagree -- on tables that HAVE been trained:

  pretrained   the reference's only real data: examples/pretrained/TransE/model.vec.pt, the WHOLE checkpoint (FB15k,
               E = 14 951, R = 1 345, d = 50), scored with the L1 norm it was trained with and with L2.  FB15k itself is not
               available offline, so the 1 024 test triples are synthetic: half uniform (h, r, t), half PLAUSIBLE -- (h, r)
               uniform and t drawn from the 50 lowest-energy tails of (h, r) under the checkpoint (float64, L1), the region a
               trained model's real test triples live in.  The fixture carries the two tables (3.3 MB: data, not source).
               -> tests/golden/ref_trained_ranks_pretrained_fb15k.npz
  c1_transe_l1, c2_complex, c3_rotate   the BASELINE shapes after tests/golden_util.TRAINED[case] epochs of training.  65 / 116 MB
               of tables cannot be committed and the reference cannot train on the GPU box, so the tables are what the drop-in
               Trainer's bit-reproducible default step path produces (tools/make_trained_tables.py wrote them to gpurun_out/);
               the fixture keeps their SHA-256, the queries (half test, half training triples) and the reference's ranks; the
               GPU test re-trains, checks the digest and ranks.     -> tests/golden/ref_trained_ranks_<case>.npz

Usage: python oracle/make_golden_trained.py pretrained | c1_transe_l1 | c2_complex | c3_rotate ..."""
import contextlib
import glob
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_fullsize as mgf  # noqa: E402  (installs the reference shim; _ranks64, _KG, CLASS)
import torch  # noqa: E402
from pykg2vec.utils.evaluator import Evaluator  # noqa: E402
from pykg2vec.data.kgcontroller import Triple  # noqa: E402
import golden_util as gu  # noqa: E402
import ref_shim  # noqa: E402

OUT = mgf.OUT
ROOT = os.path.dirname(HERE)


def reference_ranks(model_name, hp, P, E, R, queries, hr_t, tr_h):
    """(ranks [4, n] = head, tail, filtered head, filtered tail; energies of the true candidates [n, 2] = head sweep, tail sweep)
    from the reference's own Evaluator.test over a model holding the tables P."""
    n = len(queries)
    mk = lambda arr: [Triple(int(a), int(b), int(c)) for a, b, c in arr]
    cfg = types.SimpleNamespace(
        tot_entity=E, tot_relation=R, device="cpu", optimizer="sgd", learning_rate=0.01, neg_rate=hp.get("neg_rate", 1),
        alpha=hp.get("alpha", 0.1), margin=hp.get("margin", 1.0), batch_size=128, tot_train_triples=1, epochs=1000, test_num=n,
        debug=False, load_from_data=None, hits=[1, 3, 5, 10], patience=3, dataset_name="synthetic", sampling="uniform",
        knowledge_graph=mgf._KG({"triplets_train": [], "triplets_valid": mk(queries[:4]), "triplets_test": mk(queries),
                                 "hr_t": hr_t, "tr_h": tr_h}))
    for k, v in hp.items():
        setattr(cfg, k, v)
    cfg.summary = lambda: None
    mod, cls = mgf.CLASS[model_name].split(".")
    model = getattr(__import__("pykg2vec.models." + mod, fromlist=[cls]), cls)(**cfg.__dict__)
    model.load_state_dict({k + ".weight": torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).copy()) for k, v in P.items()})
    model.eval()
    ev = Evaluator(model, cfg)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        ev.test(ev.test_data, n, epoch=0)
    mc = ev.metric_calculator
    ranks = np.stack([np.asarray(x, np.int64) for x in (mc.rank_head, mc.rank_tail, mc.f_rank_head, mc.f_rank_tail)])
    with torch.no_grad():
        ents = torch.arange(E)
        st = []
        for h, r, t in queries:
            sh = model(ents, torch.full((E,), int(r)), torch.full((E,), int(t)))
            stl = model(torch.full((E,), int(h)), torch.full((E,), int(r)), ents)
            st.append((sh[int(h)].item(), stl[int(t)].item()))
    return ranks, np.asarray(st, np.float32)


def run_pretrained():
    sd = torch.load(os.path.join(ref_shim.REFERENCE_ROOT, "examples/pretrained/TransE/model.vec.pt"))
    ent = sd["ent_embeddings.weight"].numpy().astype(np.float32)
    rel = sd["rel_embeddings.weight"].numpy().astype(np.float32)
    E, R, d = ent.shape[0], rel.shape[0], ent.shape[1]
    assert (E, R, d) == (14951, 1345, 50)
    rng = np.random.default_rng(20260930)
    draw = lambda n: np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1).astype(np.int64)
    train, valid = draw(60000), draw(2000)
    n = 1024
    test = draw(n)
    # the plausible half: t among the 50 best tails of (h, r) under the checkpoint's own L1 energies (normalised rows, float64)
    e64 = ent.astype(np.float64)
    e64 /= np.maximum(np.linalg.norm(e64, axis=1, keepdims=True), 1e-12)
    r64 = rel.astype(np.float64)
    r64 /= np.maximum(np.linalg.norm(r64, axis=1, keepdims=True), 1e-12)
    for i in range(n // 2, n):
        h, r = test[i, 0], test[i, 1]
        s = np.abs(e64[h] + r64[r] - e64).sum(1)
        test[i, 2] = np.argsort(s, kind="stable")[rng.integers(50)]
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), test, R)
    P = {"ent_embeddings": ent, "rel_embeddings": rel}
    rec = {"E": np.int64(E), "R": np.int64(R), "ent_embeddings": ent, "rel_embeddings": rel, "train": train, "valid": valid,
           "test": test, "n": np.int64(n)}
    for l1 in (True, False):
        tag = "l1" if l1 else "l2"
        hp = dict(hidden_size=d, l1_flag=l1, margin=1.0)
        ranks, st = reference_ranks("transe", hp, P, E, R, test, hr_t, tr_h)
        r64k = mgf._ranks64(dict(model="transe", hp=hp), P, test, hr_t, tr_h)
        rec["ranks_" + tag], rec["ranks64_" + tag], rec["true_scores_" + tag] = ranks, r64k, st
        print("pretrained", tag, "rank entries where the reference's fp32 differs from float64: %d of %d; median tail rank uniform half %d, "
              "plausible half %d" % ((ranks != r64k).sum(), ranks.size, np.median(ranks[1, :n // 2]), np.median(ranks[1, n // 2:])),
              flush=True)
    np.savez_compressed(os.path.join(OUT, "ref_trained_ranks_pretrained_fb15k.npz"), **rec)


def run_trained(name):
    spec, P0, train, valid, test, _ids, _batch = gu.fullsize_inputs(name)
    parts = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "trained_%s_part*.npz" % name)))
    assert parts, "run tools/make_trained_tables.py %s on the GPU first" % name
    P = {}
    for f in parts:
        with np.load(f) as z:
            P.update({k: z[k] for k in z.files})
    assert sorted(P) == sorted(P0) and all(P[k].shape == P0[k].shape for k in P0), (sorted(P), sorted(P0))
    digest = gu.tables_sha256(P)
    import json
    meta = json.load(open(os.path.join(ROOT, "gpurun_out", "trained_%s_meta.json" % name)))
    assert meta["digest"] == digest, "the table files under gpurun_out/ are not the set the GPU run hashed"
    E, R, hp = spec["E"], spec["R"], dict(spec["hp"])
    hp.setdefault("margin", 1.0)
    queries = gu.trained_queries(name, train, test)
    hr_t, tr_h = gu.query_filters(np.concatenate([train, valid, test]), queries, R)
    ranks, st = reference_ranks(spec["model"], hp, P, E, R, queries, hr_t, tr_h)
    print("reference ranks done:", name, flush=True)
    r64 = mgf._ranks64(spec, P, queries, hr_t, tr_h)
    moved = float(np.mean([np.abs(P[k] - P0[k]).mean() / max(1e-12, np.abs(P0[k]).mean()) for k in P0]))
    np.savez_compressed(os.path.join(OUT, "ref_trained_ranks_%s.npz" % name), name=name, digest=digest, queries=queries, ranks=ranks,
                        ranks64=r64, true_scores=st, epochs=np.int64(meta["epochs"]), first_loss=np.float64(meta["first_loss"]),
                        last_loss=np.float64(meta["last_loss"]), step_path=meta["path"])
    n = len(queries)
    print("wrote", name, "digest", digest[:16], "mean |table - init| / mean |init| = %.2f" % moved,
          "reference vs float64: %d of %d rank entries differ;" % ((ranks != r64).sum(), ranks.size),
          "median filtered tail rank: test half %d, train half %d" % (np.median(ranks[3, :n - n // 2]), np.median(ranks[3, n - n // 2:])), flush=True)


if __name__ == "__main__":
    for arg in sys.argv[1:]:
        run_pretrained() if arg == "pretrained" else run_trained(arg)
