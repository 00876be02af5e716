"""Dev tool: the two-phase owner-computes step (KGE_PW_PULL=1) against the atomic-scatter step (KGE_PW_PULL=0: hipGraph replay in the
launch-bound regime, eager beyond) for DistMult / ComplEx over batch sizes and graph shapes; us per step."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = [("distmult", "fb15k", 100, 128, "adagrad"), ("distmult", "fb15k", 100, 4096, "adagrad"), ("distmult", "fb15k", 100, 32768, "adagrad"),
         ("distmult", "fb15k", 100, 32768, "adam"), ("complex", "wn18rr", 200, 128, "adagrad"), ("complex", "wn18rr", 200, 5000, "adagrad"),
         ("complex", "wn18rr", 200, 5000, "adam"), ("complex", "fb15k", 200, 4096, "adagrad"), ("complex", "yago310", 200, 8192, "adagrad")]
SHAPES = {"fb15k": (14951, 1345, 483142), "wn18rr": (40943, 11, 86835), "yago310": (123182, 37, 1079040)}
if len(sys.argv) > 1:
    import numpy as np, torch, hip_util
    from pykg2vec_amd.trainer import Trainer
    model, ds, d, B, opt = CASES[int(sys.argv[1])]
    E, R, NTR = SHAPES[ds]
    rng = np.random.default_rng(1234)
    train = np.stack([rng.integers(E, size=NTR), rng.integers(R, size=NTR), rng.integers(E, size=NTR)], 1)
    hp = dict(hidden_size=d, lmbda=1e-4, neg_rate=1)
    cfg = hip_util.make_config(E, R, hp, train, train[:16], train[:16], optimizer=opt, lr=0.01, batch_size=B)
    torch.manual_seed(0)
    m = hip_util.model_from_params(model, {}, hp, E, R, train=train)
    tr = Trainer(m, cfg); tr.build_model(); tr.generator = tr._new_generator()
    K_ = min(200, NTR // B)
    cfg.tot_train_triples = B * K_
    tr.train_model_epoch(0); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.train_model_epoch(1); torch.cuda.synchronize()
    print(json.dumps({"us": (time.perf_counter() - t0) / K_ * 1e6, "own": getattr(tr, "_own", None) is not None, "graph": tr._graph is not None}))
    sys.exit(0)
print("| model | graph | d | B | optimizer | atomic-scatter step us (mode) | owner-computes two-phase step us |")
print("|---|---|---|---|---|---|---|")
for i, (model, ds, d, B, opt) in enumerate(CASES):
    res = {}
    for v in ("0", "1"):
        env = dict(os.environ, KGE_PW_PULL=v)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), str(i)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res[v] = json.loads(line[-1]) if line else {"us": float("nan"), "graph": False, "err": out.stderr[-300:]}
    print("| %s | %s | %d | %d | %s | %.1f (%s) | %.1f |" % (model, ds, d, B, opt, res["0"]["us"], "graph" if res["0"]["graph"] else "eager", res["1"]["us"]), flush=True)
