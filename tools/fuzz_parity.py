"""Dev tool: randomised shape sweep of the fused train step and the rank sweep against the numpy oracle
(one MI355X).  Prints the cases that violate the parity tolerances; exits non-zero if any."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import hip_util, kge_oracle as ko
from pykg2vec_amd.trainer import Trainer
from pykg2vec_amd.evaluator import Evaluator

rng = np.random.default_rng(int(os.environ.get("SEED", "2024")))
MODELS = ["transe", "transh", "transd", "transm", "transr", "rotate", "rescal", "ntn", "distmult", "complex", "complexn3",
          "analogy", "cp", "simple", "simple_ignr", "quate"]
bad = 0
for it in range(int(os.environ.get("ITERS", "80"))):
    model = MODELS[it % len(MODELS)]
    E, R = int(rng.integers(3, 400)), int(rng.integers(1, 40))
    B = int(rng.integers(1, 300))
    neg = int(rng.integers(1, 6)) if (model in ko.POINTWISE or model == "rotate") else 1
    d = int(rng.integers(2, 140))  # 1-dim rows normalise to +-1: every score ties, ranks are then unspecified
    if model == "transh":
        d = max(d, 3)  # in 2-D the hyperplane projection leaves a 1-dim row: the same degenerate normalisation
    if model == "quate":
        E = max(E, R)  # the reference's QuatE relation tables have tot_entity rows (pointwise.py:653-657): it needs E >= R
    hp = dict(l1_flag=bool(rng.integers(2)), margin=float(rng.uniform(0.5, 8)), lmbda=float(rng.uniform(0, 0.2)), alpha=float(rng.uniform(0.2, 2)))
    if model in ("transd",):
        hp.update(ent_hidden_size=d, rel_hidden_size=d)
    elif model == "transr":
        hp.update(ent_hidden_size=min(d, 128), rel_hidden_size=int(rng.integers(2, 129)))
    elif model == "ntn":
        hp.update(ent_hidden_size=max(2, min(d, 64)), rel_hidden_size=int(rng.integers(2, 48)))
    elif model == "analogy":
        hp.update(hidden_size=2 * max(1, d // 2))
    elif model == "rescal":
        hp.update(hidden_size=min(d, 96))
    else:
        hp.update(hidden_size=d)
    shape_kw = {k: v for k, v in hp.items() if k in ("hidden_size", "ent_hidden_size", "rel_hidden_size")}
    if model == "rotate":
        shape_kw["margin"] = hp["margin"]
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, **shape_kw)
    pos = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
    nh = np.repeat(pos[:, 0], neg); nr = np.repeat(pos[:, 1], neg); nt = np.repeat(pos[:, 2], neg)
    flip = rng.random(B * neg) > 0.5
    rnd = rng.integers(E, size=B * neg)
    nh = np.where(flip, nh, rnd); nt = np.where(flip, rnd, nt)
    pointwise = model in ko.POINTWISE
    batch = ko.pointwise_layout(pos, nh, nr, nt, neg) if pointwise else (pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt)
    hp_run = dict(hp, neg_rate=neg)
    if model == "transm":
        hp_run["theta"] = ko.transm_theta(pos, R)
    try:
        loss_ref, G_ref, sc, Pn = ko.train_step_grads(model, P, batch, **hp_run)
        m = hip_util.model_from_params(model, P, {k: v for k, v in hp.items()}, E, R, train=pos)
        cfg = hip_util.make_config(E, R, hp_run, pos, pos[:1], pos[:min(B, 8)])
        tr = Trainer(m, cfg, use_graph=False); tr.build_model()
        b = [hip_util.dev(x) for x in batch]
        loss = (tr.train_step_pointwise(*b) if pointwise else tr.train_step_pairwise(*b)).item()
        ok = np.isclose(loss, loss_ref, rtol=1e-4, atol=1e-4)
        worst = 0.0
        for (name, _), g in zip(hip_util.table_parameters(m), tr.flat.grad_views):
            ref = G_ref[name.split(".")[0]]
            scale = max(1e-3, float(np.abs(ref).max()))
            err = float(np.abs(g.cpu().numpy() - ref).max()) / scale
            worst = max(worst, err)
        ok = ok and worst < 2e-3
        # ranks of a few triples against the oracle (band of +-2 for fp32 near-ties)
        if model != "rescal":
            Pe = P
        else:
            Pe = ko.rescal_normalize_tables(P)
        tr.flat.grad.zero_()
        ev = Evaluator(m, cfg)
        n = min(B, 8)
        ranks = ev.rank_all(pos, n).cpu().numpy()
        hr_t, tr_h = cfg.knowledge_graph.cache["hr_t"], cfg.knowledge_graph.cache["tr_h"]
        _, ref = ko.evaluate(model, P, pos[:n], hr_t, tr_h, **hp_run)
        refm = np.stack([ref["head"], ref["tail"], ref["fhead"], ref["ftail"]])
        rank_ok = np.abs(ranks - refm).max() <= 2 and (ranks != refm).mean() < 0.2
        if not (ok and rank_ok):
            bad += 1
            print("FAIL", model, dict(E=E, R=R, B=B, neg=neg), {k: v for k, v in hp.items() if "size" in k}, "loss", loss, loss_ref,
                  "grad err", worst, "rank diff", int(np.abs(ranks - refm).max()), flush=True)
    except Exception as ex:  # noqa
        bad += 1
        print("ERROR", model, dict(E=E, R=R, B=B, neg=neg), {k: v for k, v in hp.items() if "size" in k}, repr(ex)[:300], flush=True)
print("fuzz done: %d cases, %d bad" % (it + 1, bad))
sys.exit(1 if bad else 0)
