"""Dev tool: randomised check of the round-3 training paths against the atomic-scatter paths on the sampler's batches (same seed
=> same batches and Philox draws): the staged pointwise owner-computes step (DistMult / ComplEx / ComplexN3), the TransH / TransD
gradients without atomics, the two-launch TransE / TransM step (must be BIT-identical to the one-launch step for L1), and the
one-launch RESCAL pair step (incl. odd-vector hidden sizes).  Random graph / batch / row sizes, tiny entity sets (bucket overflow),
few relations (rows cut across workgroups), short epochs.  One MI355X."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import hip_util, kge_oracle as ko
from pykg2vec_amd.trainer import Trainer

rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
N = int(os.environ.get("ITERS", "96"))
bad = 0
counts = {}
FAMILIES = ["pointwise", "transx", "two_phase", "rescal"]
for it in range(N):
    fam = FAMILIES[it % 4]
    kind = (it // 4) % 3
    E = int(rng.integers(8, 40)) if kind == 0 else int(rng.integers(100, 4000))
    R = int(rng.integers(1, 4)) if kind == 1 else int(rng.integers(2, 60))
    B = int(rng.integers(16, 1500))
    nb = int(rng.integers(2, 4))
    n_train = nb * B + int(rng.integers(0, B))
    opt = ["sgd", "adam", "adagrad", "rms"][int(rng.integers(4))]
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    if len({tuple(x) for x in train}) > 0.4 * E * E * R:      # (the rejection loop of a saturated tiny graph would spin)
        continue
    exact = False
    if fam == "pointwise":
        model = ["distmult", "complex", "complexn3", "analogy", "cp", "simple", "simple_ignr", "quate"][int(rng.integers(8))]
        generic = model in ("analogy", "cp", "simple", "simple_ignr", "quate")      # csrc/kge_ownx.hip: any hidden size <= 256
        if model == "quate":
            E = max(E, R)      # the reference's QuatE relation tables have tot_entity rows (pointwise.py:653-657): it needs E >= R
            train[:, 0] = rng.integers(E, size=n_train); train[:, 2] = rng.integers(E, size=n_train)
        d = (2 * int(rng.integers(1, 100)) if generic else 4 * int(rng.integers(1, 80)))
        hp = dict(hidden_size=d, lmbda=float(rng.choice([0.0, 1e-3, 0.05])), neg_rate=1)
        kw, env = dict(hidden_size=d), ("KGE_PW_PULL", "0", "1")
    elif fam == "transx":
        model = ["transh", "transd"][int(rng.integers(2))]
        d = 4 * int(rng.integers(1, 70))
        kw = dict(hidden_size=d) if model == "transh" else dict(ent_hidden_size=d, rel_hidden_size=d)
        hp = dict(kw, l1_flag=bool(rng.integers(2)), margin=float(rng.uniform(0.5, 3)), neg_rate=1)
        env = ("KGE_TRANSX_OWN", "0", "1")
    elif fam == "two_phase":
        model = ["transe", "transm"][int(rng.integers(2))]
        d = 4 * int(rng.integers(1, 60))
        l1 = bool(rng.integers(4))        # mostly L1: there the two forms must agree bit for bit
        kw = dict(hidden_size=d)
        hp = dict(hidden_size=d, l1_flag=l1, margin=float(rng.uniform(0.5, 3)), neg_rate=1)
        env = ("KGE_PULL_DIR", "0", "1")
        exact = l1 and model == "transe"     # (TransM scales every term by theta_r: not exact sums, grouping matters at rounding level)
        os.environ["KGE_PULL"] = "1"
    else:
        model = "rescal"
        d = 2 * int(rng.integers(1, 60))
        kw = dict(hidden_size=d)
        hp = dict(hidden_size=d, margin=float(rng.uniform(0.5, 2)), neg_rate=1)
        env = ("KGE_RESCAL_FUSED", "0", "1")      # separate renormalisation pass vs folded into the row-owner optimiser + bitmaps
    P = ko.init_params("transe" if model == "transm" else model, rng, tot_entity=E, tot_relation=R, **kw)
    if os.environ.get("ONLY_IT") and it != int(os.environ["ONLY_IT"]):      # replay one case (same random stream), e.g. with FORCE_OPT
        continue
    opt = os.environ.get("FORCE_OPT", opt)
    res = {}
    for val in env[1:]:
        os.environ[env[0]] = val
        cfg = hip_util.make_config(E, R, hp, train, train[:2], train[:2], optimizer=opt, lr=float(os.environ.get("FORCE_LR", 0.01)), batch_size=B)
        m = hip_util.model_from_params(model, P, hp, E, R, train=train)
        tr = Trainer(m, cfg, use_graph=False)
        tr.build_model()
        tr.generator = tr._new_generator()
        losses = [tr.train_model_epoch(e) for e in range(int(os.environ.get("FORCE_EPOCHS", 2)))]
        res[val] = (losses, [p.detach().cpu().numpy().copy() for _, p in hip_util.table_parameters(m)])
        del tr, m
    os.environ.pop(env[0], None)
    os.environ.pop("KGE_PULL", None)
    counts[fam] = counts.get(fam, 0) + 1
    if os.environ.get("ONLY_IT"):      # replay: how far the two paths' tables are apart, relative to how far they moved from the start
        init = [np.asarray(P[k]) for k in ko.PARAM_NAMES["transe" if model == "transm" else model]]
        for nme, a, b, p0 in zip(ko.PARAM_NAMES["transe" if model == "transm" else model], res["0"][1], res["1"][1], init):
            moved = np.abs(a - p0)
            rel = np.abs(a - b) / np.maximum(moved, 1e-12)
            print("   ", nme, "max |a-b| %.3g" % np.abs(a - b).max(), "median move %.3g" % np.median(moved), "median / 99th pct of |a-b| / move: %.3g / %.3g" % (np.median(rel), np.percentile(rel, 99)), flush=True)
    # (rms: the atomic arm differs from itself run to run by up to ~1.5e-4 of the second epoch's loss on tables of two relations)
    ok = np.allclose(res["0"][0], res["1"][0], rtol=3e-4 if opt == "rms" else 1e-4)
    fracs = []
    for a, b in zip(res["0"][1], res["1"][1]):
        if exact and fam == "two_phase" and (E + R) * 1 > 0:
            # (the two-phase form cuts items at 32 incidences, the one-phase form at 8: L1 sums are exact in any grouping)
            ok = ok and np.array_equal(a, b)
        else:
            frac = (~np.isclose(a, b, atol=3e-5, rtol=1e-4)).mean()
            fracs.append(round(float(frac), 5))
            # (L1 distances: a residual element within rounding of zero can take either sign on the two paths -- their group
            # reductions add in different orders -- which moves one parameter element by 2 lr: isolated entries, also under SGD)
            l1_model = bool(hp.get("l1_flag", False))
            # RMSprop divides by sqrt(0.01 g^2 + ...): an element's update is ~10 lr g / |g|_recent, so a RELATIVE difference of 3e-4
            # between the two paths' gradient sums already moves a parameter by the tolerance -- and the normal-vector / relation
            # gradients of an L1 model are sums of hundreds of cancelling terms whose fp32 value depends on the summation tree at
            # that level (the same case agrees under SGD / Adam / Adagrad: replay with ONLY_IT / FORCE_OPT).  Under rms the
            # element-wise check therefore only bounds the damage; the losses (rtol 3e-4 under rms, 1e-4 otherwise) carry the comparison.
            lim = (2e-3 if l1_model else 0.0) if opt == "sgd" else (1.0 if a.size < 4096 else 0.1) if opt == "rms" else 1e-2
            ok = ok and frac <= lim
    if not ok:
        bad += 1
        print("MISMATCH", "it=%d" % it, fam, model, dict(E=E, R=R, B=B, d=d, n_train=n_train, opt=opt, hp=hp), res["0"][0], res["1"][0], "differing fraction per table", fracs, flush=True)
print(f"own fuzz done: {sum(counts.values())} cases {counts}, {bad} bad")
sys.exit(1 if bad else 0)
