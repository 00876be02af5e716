"""Dev tool: replay of ONE fuzz_pull case (SEED, ONLY_IT) step by step: both arms against the numpy oracle after the first step, the
entries that are off by a whole L1 sign unit and the residual elements behind them.  One MI355X."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import hip_util, kge_oracle as ko
from pykg2vec_amd import kernels as K
from pykg2vec_amd.trainer import Trainer
SEED = int(os.environ.get("SEED", "1003")); IT = int(os.environ.get("ONLY_IT", "69"))
rng = np.random.default_rng(SEED)
for it in range(IT + 1):
    model = "transm" if it % 3 == 2 else "transe"
    kind = it % 4
    E = int(rng.integers(8, 60)) if kind == 0 else int(rng.integers(100, 5000))
    R = int(rng.integers(1, 50))
    B = int(rng.integers(8, 600)) if kind != 3 else int(rng.integers(4, 24))
    d = 4 * int(rng.integers(1, 40))
    nb = int(rng.integers(2, 6))
    n_train = nb * B + int(rng.integers(0, B))
    opt = ["sgd", "adam", "adagrad", "rms"][int(rng.integers(4))]
    l1 = bool(rng.integers(2))
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    if len({tuple(x) for x in train}) > 0.5 * E * E * R:
        continue
    hp = dict(hidden_size=d, l1_flag=l1, margin=float(rng.uniform(0.5, 4)))
    P = ko.init_params("transe", rng, tot_entity=E, tot_relation=R, hidden_size=d)
    if it != IT: continue
    print(dict(E=E,R=R,B=B,d=d,n_train=n_train,opt=opt,l1=l1,hp=hp))
    nsteps = (n_train + B - 1)//B
    for steps in range(1, 2):
        res = {}
        for pull in ("0", "1"):
            os.environ["KGE_PULL"] = pull
            cfg = hip_util.make_config(E, R, hp, train, train[:2], train[:2], optimizer=opt, lr=0.01, batch_size=B)
            m = hip_util.model_from_params(model, P, hp, E, R, train=train)
            tr = Trainer(m, cfg, use_graph=False)
            tr.build_model()
            tr.generator = tr._new_generator()
            cfg.tot_train_triples = min(steps * B, n_train)
            loss = tr.train_model_epoch(0)
            gen = tr.generator
            res[pull] = (loss, [p.detach().cpu().numpy().copy() for _, p in hip_util.table_parameters(m)])
            if steps == 1:
                batch = K.sample_batch(gen.triples, gen.perm, 0, B, 1, E, None, gen.slots, gen.seed, 0)
                nbt = tuple(a.cpu().numpy() for a in batch)
                loss_ref, G_ref, _, _ = ko.train_step_grads("transe", P, nbt, l1_flag=l1, margin=hp["margin"])
                Pn = {k: v.copy() for k, v in P.items()}
                st = ko.optimizer_init(opt, Pn); ko.optimizer_step(opt, Pn, G_ref, st, 0.01)
                print(" arm", pull, "loss", loss, "oracle", loss_ref)
                for k, p in hip_util.table_parameters(m):
                    got, ref = p.detach().cpu().numpy(), Pn[k.split(".")[0]]
                    print("   ", k, "max|got-ref|", np.abs(got-ref).max(), "frac>3e-5", (np.abs(got-ref)>3e-5).mean())
                    rr, cc = np.nonzero(np.abs(got-ref)>3e-5)
                    for row in sorted(set(rr.tolist())):
                        cols = cc[rr==row]
                        big = cols[np.abs((got-ref)[row, cols]) > 1e-3]
                        print("      row", row, "entries off", len(cols), "of which by a whole sign unit (lr / norm):", big.tolist(), (got-ref)[row, big].round(5).tolist())
                        if k.startswith("ent"):
                            h_, r_, t_, nh_, nr_, nt_ = [np.asarray(x) for x in nbt[:6]] if len(nbt) >= 6 else (None,)*6
                            if h_ is not None:
                                for i in range(len(h_)):
                                    if row in (h_[i], t_[i], nh_[i], nt_[i]) and len(big):
                                        def resid(a, b, c):   # float64 residual of the NORMALISED rows at the columns in question
                                            n = lambda x: x.astype(np.float64) / np.linalg.norm(x.astype(np.float64))
                                            return (n(P["ent_embeddings"][a]) + n(P["rel_embeddings"][b]) - n(P["ent_embeddings"][c]))[big]
                                        u = np.concatenate([resid(h_[i], r_[i], t_[i]), resid(nh_[i], nr_[i], nt_[i])])
                                        if np.abs(u).min() < 1e-6:
                                            print("         pair", i, (int(h_[i]), int(r_[i]), int(t_[i])), (int(nh_[i]), int(nr_[i]), int(nt_[i])), "float64 residuals there:", u.tolist(), "<- within fp32 rounding of zero: its sign is the rounding's")
            del tr, m
        a, b = res["0"], res["1"]
        print("steps", steps, "loss", a[0], b[0], "max diff", [float(np.abs(x-y).max()) for x, y in zip(a[1], b[1])], "rows differing", [int((np.abs(x-y).max(1)>3e-5).sum()) for x, y in zip(a[1], b[1])])
