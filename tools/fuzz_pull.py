"""Dev tool: randomised check that the owner-computes TransE / TransM step (the default training path) reproduces the
atomic-scatter path: same batches, same negatives, same weights up to fp32 summation order, over random graph / batch /
row sizes (explicit and compact incidence index, bucket overflow on tiny entity sets, short epochs).  One MI355X."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import hip_util, kge_oracle as ko
from pykg2vec_amd import kernels as K
from pykg2vec_amd.trainer import Trainer

rng = np.random.default_rng(int(os.environ.get("SEED", "11")))
N = int(os.environ.get("ITERS", "48"))
bad = compact = 0
for it in range(N):
    model = "transm" if it % 3 == 2 else "transe"
    kind = it % 4
    E = int(rng.integers(8, 60)) if kind == 0 else int(rng.integers(100, 5000))        # kind 0: every entity drawn many times
    R = int(rng.integers(1, 50))
    B = int(rng.integers(8, 600)) if kind != 3 else int(rng.integers(4, 24))            # kind 3: batches that touch few rows
    d = 4 * int(rng.integers(1, 40))
    nb = int(rng.integers(2, 6))
    n_train = nb * B + int(rng.integers(0, B))
    opt = ["sgd", "adam", "adagrad", "rms"][int(rng.integers(4))]
    l1 = bool(rng.integers(2))
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    # (a tiny graph whose every corruption is a train triple would make the reference's rejection loop spin as well)
    if len({tuple(x) for x in train}) > 0.5 * E * E * R:
        continue
    hp = dict(hidden_size=d, l1_flag=l1, margin=float(rng.uniform(0.5, 4)))
    P = ko.init_params("transe", rng, tot_entity=E, tot_relation=R, hidden_size=d)
    if os.environ.get("ONLY_IT") and it != int(os.environ["ONLY_IT"]):   # replay of one case: the generator is advanced as usual
        continue
    res = {}
    for pull in os.environ.get("ARMS", "0,1").split(","):   # (ARMS=0,0: the atomic arm against ITSELF -- its own run-to-run spread)
        os.environ["KGE_PULL"] = pull
        cfg = hip_util.make_config(E, R, hp, train, train[:2], train[:2], optimizer=opt, lr=0.01, batch_size=B)
        m = hip_util.model_from_params(model, P, hp, E, R, train=train)
        tr = Trainer(m, cfg, use_graph=False)
        tr.build_model()
        tr.generator = tr._new_generator()
        losses = [tr.train_model_epoch(e) for e in range(2)]
        if pull == "1":
            assert tr._pull is not None
            compact += int(tr.generator.pull_index().compact)
        res.setdefault("first" if "first" not in res else "second", (losses, [p.detach().cpu().numpy().copy() for _, p in hip_util.table_parameters(m)]))
        del tr, m
    r0, r1 = res["first"], res["second"]
    # (rms: a residual element within rounding of zero takes either sign on the two paths -- they normalise in a different order -- and
    # RMSprop turns that +-2 of gradient into a full 10 lr step of the element: isolated entries, but the SECOND epoch's loss moves by
    # up to ~1e-3 relative; each arm agrees with itself run to run: ONLY_IT=<it> ARMS=0,0 / 1,1)
    # Adam / Adagrad do the same with a full lr step (second-epoch loss up to ~1e-4 apart at d = 4, where an element is a quarter of a row)
    ok = np.allclose(r0[0], r1[0], rtol=5e-3 if opt == "rms" else (5e-5 if opt == "sgd" else 2e-4))
    fracs = []
    for a, b in zip(r0[1], r1[1]):
        frac = (~np.isclose(a, b, atol=3e-5, rtol=1e-4)).mean()
        fracs.append(round(float(frac), 5))
        # (isolated entries: at most 0.5 % of a table or 16 entries of a small one -- a one-relation table has 56; under rms, where every
        #  such entry moves by 10 lr, the element-wise check only bounds the damage, as in fuzz_own.py, and the losses carry the comparison)
        lim = 0.0 if opt == "sgd" else ((1.0 if a.size < 4096 else 0.1) if opt == "rms" else max(5e-3, 16.0 / a.size))
        ok = ok and frac <= lim
    if not ok:
        bad += 1
        print("MISMATCH it=%d" % it, model, dict(E=E, R=R, B=B, d=d, n_train=n_train, opt=opt, l1=l1), r0[0], r1[0], "differing fraction per table", fracs, flush=True)
print(f"pull fuzz done: {N} cases, {bad} bad; compact index in {compact}")
sys.exit(1 if bad else 0)
