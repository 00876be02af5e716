"""Dev tool: randomised shape sweep of the matrix-core rank sweep (k_eval_gemm + k_eval_prepare4 / k_eval_prepare +
k_eval_target_filter_chain) on one MI355X: ranks must be EXACT functions of the sweep's own energies (oracle
rank_from_scores on the GPU score rows), energies within fp32 tolerance of the VALU sweep's.  Random models, entity counts,
widths (incl. non-multiples of 4), query counts below / above the 512-query dispatch rule (forced on), filter-list lengths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import hip_util, kge_oracle as ko
from pykg2vec_amd import kernels as K
from pykg2vec_amd.evaluator import Evaluator

rng = np.random.default_rng(int(os.environ.get("SEED", "31")))
MODELS = ["distmult", "complex", "complexn3", "analogy", "rescal", "cp", "simple", "simple_ignr", "quate", "rotate"]
bad = 0
longest_seen, odd_width, queries = 0, 0, 0
for it in range(int(os.environ.get("ITERS", "60"))):
    model = MODELS[it % len(MODELS)]
    E, R = int(rng.integers(65, 2500)), int(rng.integers(1, 12))
    n = int(rng.integers(3, 700))
    d = int(rng.integers(2, 150))
    if model == "analogy":
        d = 2 * max(1, d // 2)
    if model == "rescal":
        d = min(d, 64)
    if model == "quate":
        E = max(E, R)
    hp = dict(hidden_size=d, lmbda=0.01, margin=float(rng.uniform(2, 12)), alpha=1.0)
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, hidden_size=d, margin=hp["margin"])
    nk = int(rng.integers(n, 6 * n + 50))
    # a few heavy (h, r) / (t, r) groups so that filter lists of every length class occur
    heads = rng.integers(E, size=nk); heads[: nk // 3] = rng.integers(min(E, 4), size=nk // 3)
    known = np.stack([heads, rng.integers(R, size=nk), rng.integers(E, size=nk)], 1)
    test = known[rng.permutation(nk)[:n]]
    try:
        cfg = hip_util.make_config(E, R, hp, known, known[:4], test)
        m = hip_util.model_from_params(model, P, hp, E, R)
        if model == "rescal":
            m.normalize_tables()
        os.environ["KGE_EVAL_GEMM"] = "1"
        ranks = Evaluator(m, cfg).rank_all(test, n).cpu().numpy()
        scores = K.eval_sweep_scores(m.make_desc(), hip_util.dev(test)).cpu().numpy()
        os.environ["KGE_EVAL_GEMM"] = "0"
        scores_v = K.eval_sweep_scores(m.make_desc(), hip_util.dev(test)).cpu().numpy()
        hr_t, tr_h = cfg.knowledge_graph.cache["hr_t"], cfg.knowledge_graph.cache["tr_h"]
        wrong = 0
        for i, (h, r, t) in enumerate(test):
            wrong += (ranks[1, i], ranks[3, i]) != ko.rank_from_scores(scores[2 * i], int(t), hr_t[(int(h), int(r))])
            wrong += (ranks[0, i], ranks[2, i]) != ko.rank_from_scores(scores[2 * i + 1], int(h), tr_h[(int(t), int(r))])
        close = np.allclose(scores, scores_v, atol=2e-5, rtol=2e-5)
        longest_seen = max(longest_seen, max(len(v) for v in hr_t.values())); odd_width += d % 4 != 0; queries += 2 * n
        if wrong or not close:
            bad += 1
            print("FAIL", model, dict(E=E, R=R, n=n, d=d, known=nk, longest=max(len(v) for v in hr_t.values())), "wrong ranks", wrong,
                  "score err", float(np.abs(scores - scores_v).max()), flush=True)
    except Exception as ex:  # noqa
        bad += 1
        print("ERROR", model, dict(E=E, R=R, n=n, d=d), repr(ex)[:300], flush=True)
print("fuzz done: %d cases, %d bad (%d queries checked, longest filter list %d, %d cases with a width that is no multiple of 4)" % (it + 1, bad, queries, longest_seen, odd_width))
sys.exit(1 if bad else 0)
