"""Dev tool: step time / eval throughput of the BASELINE.json configs (synthetic dataset-shaped data) on one MI355X."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_util
from pykg2vec_amd.trainer import Trainer
from pykg2vec_amd.evaluator import Evaluator

SHAPES = {"umls": (135, 46, 5216, 661), "fb15k": (14951, 1345, 483142, 59071), "wn18rr": (40943, 11, 86835, 3134),
          "fb15k237": (14541, 237, 272115, 20466), "yago310": (123182, 37, 1079040, 5000)}
CONFIGS = [
    ("C0 TransE UMLS d=50 B=128 adam", "transe", "umls", dict(hidden_size=50, l1_flag=True, margin=0.8), "adam", 128, 1, 661),
    ("C1 TransE FB15k d=100 B=128 adam", "transe", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 128, 1, 0),
    ("C1 TransE FB15k d=100 B=4096 adam", "transe", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 4096, 1, 0),
    ("C1 TransE FB15k d=100 B=8192 adam", "transe", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 8192, 1, 0),
    ("C1 TransE FB15k d=100 B=16384 adam", "transe", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 16384, 1, 0),
    ("C1 TransE FB15k d=100 B=32768 adam", "transe", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 32768, 1, 8192),
    ("C1 TransE-L2 FB15k d=100 B=32768 sgd", "transe", "fb15k", dict(hidden_size=100, l1_flag=False, margin=1.0), "sgd", 32768, 1, 8192),
    ("C1 TransE-L2 FB15k d=100 B=32768 adam", "transe", "fb15k", dict(hidden_size=100, l1_flag=False, margin=1.0), "adam", 32768, 1, 0),
    ("C1 TransE-L2 FB15k d=100 B=8192 adam", "transe", "fb15k", dict(hidden_size=100, l1_flag=False, margin=1.0), "adam", 8192, 1, 0),
    ("TransM-L2 FB15k d=100 B=32768 adam", "transm", "fb15k", dict(hidden_size=100, l1_flag=False, margin=1.0), "adam", 32768, 1, 0),
    ("TransH FB15k d=100 B=8192 adam", "transh", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 8192, 1, 0),
    ("TransD FB15k d=100 B=8192 adam", "transd", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=True, margin=1.0), "adam", 8192, 1, 0),
    ("DistMult FB15k d=100 B=128 adam", "distmult", "fb15k", dict(hidden_size=100, lmbda=1e-4), "adam", 128, 1, 0),
    ("ComplEx WN18RR d=200 B=128 adam", "complex", "wn18rr", dict(hidden_size=200, lmbda=1e-4), "adam", 128, 1, 0),
    ("ComplEx WN18RR d=200 B=1024 adam", "complex", "wn18rr", dict(hidden_size=200, lmbda=1e-4), "adam", 1024, 1, 0),
    ("DistMult FB15k d=100 B=128 adagrad", "distmult", "fb15k", dict(hidden_size=100, lmbda=1e-4), "adagrad", 128, 1, 0),
    ("ComplEx WN18RR d=200 B=128 adagrad", "complex", "wn18rr", dict(hidden_size=200, lmbda=1e-4), "adagrad", 128, 1, 0),
    ("ComplEx WN18RR d=200 B=512 adagrad", "complex", "wn18rr", dict(hidden_size=200, lmbda=1e-4), "adagrad", 512, 1, 0),
    ("TransH FB15k d=100 B=128 adam", "transh", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 128, 1, 0),
    ("TransH FB15k d=100 B=1024 adam", "transh", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 1024, 1, 0),
    ("TransD FB15k d=100 B=1024 adam", "transd", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=True, margin=1.0), "adam", 1024, 1, 0),
    ("DistMult FB15k d=100 B=1024 adagrad", "distmult", "fb15k", dict(hidden_size=100, lmbda=1e-4), "adagrad", 1024, 1, 0),
    ("DistMult FB15k d=100 B=4096 adagrad", "distmult", "fb15k", dict(hidden_size=100, lmbda=1e-4), "adagrad", 4096, 1, 0),
    ("ComplEx FB15k d=200 B=4096 adagrad", "complex", "fb15k", dict(hidden_size=200, lmbda=1e-4), "adagrad", 4096, 1, 0),
    ("ComplEx FB15k d=200 B=4096 adam", "complex", "fb15k", dict(hidden_size=200, lmbda=1e-4), "adam", 4096, 1, 0),
    ("TransH FB15k d=100 B=4096 adam", "transh", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 4096, 1, 0),
    ("TransD FB15k d=100 B=4096 adam", "transd", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=True, margin=1.0), "adam", 4096, 1, 0),
    ("DistMult FB15k d=100 B=8192 adagrad", "distmult", "fb15k", dict(hidden_size=100, lmbda=1e-4), "adagrad", 8192, 1, 0),
    ("TransH FB15k d=100 B=32768 adam", "transh", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 32768, 1, 2048),
    ("TransD FB15k d=100 B=32768 adam", "transd", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=True, margin=1.0), "adam", 32768, 1, 2048),
    ("DistMult FB15k d=100 B=32768 adagrad", "distmult", "fb15k", dict(hidden_size=100, lmbda=1e-4), "adagrad", 32768, 1, 8192),
    ("C2 ComplEx WN18RR d=200 B=5000 adagrad", "complex", "wn18rr", dict(hidden_size=200, lmbda=1e-4), "adagrad", 5000, 1, 3134),
    ("ANALOGY FB15k d=200 B=4096 adagrad", "analogy", "fb15k", dict(hidden_size=200, lmbda=1e-4), "adagrad", 4096, 1, 2048),
    ("C3 RotatE FB15k-237 d=1000 B=1024 neg16 adam", "rotate", "fb15k237", dict(hidden_size=1000, margin=24.0, neg_rate=16, alpha=1.0), "adam", 1024, 16, 2048),
    ("C4 RESCAL YAGO3-10 k=200 B=1024 adam", "rescal", "yago310", dict(hidden_size=200, margin=1.0), "adam", 1024, 1, 1024),
    ("RESCAL FB15k k=50 B=128 adam (preset)", "rescal", "fb15k", dict(hidden_size=50, margin=1.0), "adam", 128, 1, 1024),
    ("TransM FB15k d=50 B=1200 sgd (preset)", "transm", "fb15k", dict(hidden_size=50, l1_flag=False, margin=0.5), "sgd", 1200, 1, 8192),
    ("TransM FB15k d=100 B=32768 adam", "transm", "fb15k", dict(hidden_size=100, l1_flag=True, margin=1.0), "adam", 32768, 1, 8192),
    ("TransR FB15k 50/50 B=4800 sgd (preset)", "transr", "fb15k", dict(ent_hidden_size=50, rel_hidden_size=50, l1_flag=True, margin=1.0), "sgd", 4800, 1, 2048),
    ("TransR WN18RR 100/100 B=4096 adam", "transr", "wn18rr", dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=False, margin=1.0), "adam", 4096, 1, 1024),
    ("CP FB15k d=50 B=128 adagrad (preset)", "cp", "fb15k", dict(hidden_size=50, lmbda=1e-4), "adagrad", 128, 1, 8192),
    ("CP FB15k d=100 B=32768 adagrad", "cp", "fb15k", dict(hidden_size=100, lmbda=1e-4), "adagrad", 32768, 1, 0),
    ("SimplE FB15k d=100 B=128 adagrad (preset)", "simple", "fb15k", dict(hidden_size=100, lmbda=0.1), "adagrad", 128, 1, 8192),
    ("SimplE FB15k d=100 B=32768 adagrad", "simple", "fb15k", dict(hidden_size=100, lmbda=0.1), "adagrad", 32768, 1, 0),
    ("SimplE_ignr FB15k d=100 B=32768 adagrad", "simple_ignr", "fb15k", dict(hidden_size=100, lmbda=0.1), "adagrad", 32768, 1, 2048),
    ("QuatE FB15k d=200 B=100 adagrad (preset)", "quate", "fb15k", dict(hidden_size=200, lmbda=0.1), "adagrad", 100, 1, 2048),
    ("QuatE FB15k d=100 B=32768 adagrad", "quate", "fb15k", dict(hidden_size=100, lmbda=0.1), "adagrad", 32768, 1, 0),
    ("QuatE WN18 d=300 B=4096 adagrad", "quate", "wn18rr", dict(hidden_size=300, lmbda=0.05), "adagrad", 4096, 1, 0),
    ("mfma-batch RESCAL YAGO3-10 k=200 B=32768 adam", "rescal", "yago310", dict(hidden_size=200, margin=1.0), "adam", 32768, 1, 0),
    ("mfma-batch RESCAL FB15k k=200 B=32768 adam", "rescal", "fb15k", dict(hidden_size=200, margin=1.0), "adam", 32768, 1, 0),
    ("mfma-batch TransR FB15k 100/100 B=32768 adam", "transr", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=True, margin=1.0), "adam", 32768, 1, 0),
    ("mfma-batch NTN FB15k d=k=100 B=32768 adam", "ntn", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0), "adam", 32768, 1, 0),
    ("ntn-mid NTN FB15k d=k=100 B=512 adam", "ntn", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0), "adam", 512, 1, 0),
    ("ntn-mid NTN FB15k d=k=100 B=1024 adam", "ntn", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0), "adam", 1024, 1, 0),
    ("ntn-mid NTN FB15k d=k=100 B=2048 adam", "ntn", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0), "adam", 2048, 1, 0),
    ("ntn-mid NTN FB15k d=k=100 B=4096 adam", "ntn", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0), "adam", 4096, 1, 0),
    ("ntn-mid NTN FB15k d=k=100 B=8192 adam", "ntn", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0), "adam", 8192, 1, 0),
    ("NTN FB15k d=k=100 B=128 adam (preset)", "ntn", "fb15k", dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0), "adam", 128, 1, 64),
]
if os.environ.get("GRAPH_UNROLL"):
    Trainer.GRAPH_UNROLL = int(os.environ["GRAPH_UNROLL"])
only = os.environ.get("ONLY")
ZIPF = os.environ.get("ZIPF_REL") == "1"   # relation ids of the TEST triples Zipf(1)-skewed, as in real test splits
rng = np.random.default_rng(1234)
for name, model, ds, hp, opt, B, neg, n_eval in CONFIGS:
    if only and only not in name:
        continue
    E, R, NTR, NTE = SHAPES[ds]
    train = np.stack([rng.integers(E, size=NTR), rng.integers(R, size=NTR), rng.integers(E, size=NTR)], 1)
    nt_ = max(int(os.environ.get("N_EVAL", n_eval)) if n_eval else 0, 16)
    rel_t = rng.integers(R, size=nt_)
    if ZIPF:
        w = 1.0 / np.arange(1, R + 1); rel_t = rng.choice(R, size=nt_, p=w / w.sum())
    test = np.stack([rng.integers(E, size=nt_), rel_t, rng.integers(E, size=nt_)], 1)
    hp2 = dict(hp); hp2.setdefault("margin", 1.0); hp2["neg_rate"] = neg
    cfg = hip_util.make_config(E, R, hp2, train, test[:16], test, optimizer=opt, lr=0.01, batch_size=B)
    cfg.hr_dummy = None
    torch.manual_seed(0)
    m = hip_util.model_from_params(model, {}, hp, E, R, train=train)
    tr = Trainer(m, cfg, use_graph=(True if os.environ.get('FORCE_GRAPH') == '1' else None)); tr.build_model()
    tr.generator = tr._new_generator()
    K_ = min(200, max(1, NTR // B))
    cfg.tot_train_triples = B * K_            # one "epoch" = K_ steps through Trainer.train_model_epoch
    tr.train_model_epoch(0)                   # warm-up (and hipGraph capture when the step is launch-bound)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_model_epoch(1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K_
    line = f"{name}: {'graph' if tr._graph is not None else 'eager'} step {dt*1e6:.1f} us -> {B*(1+neg)/dt/1e6:.2f} M scored triples/s"
    if os.environ.get("N_EVAL") and n_eval:
        n_eval = int(os.environ["N_EVAL"])
    if n_eval:
        ev = Evaluator(m, cfg)
        if os.environ.get("GROUPED_MIN"):
            ev.GROUPED_MIN_TRIPLES_PER_RELATION = float(os.environ["GROUPED_MIN"])
        ev.rank_all(test, n_eval); torch.cuda.synchronize()
        t0 = time.perf_counter(); ev.rank_all(test, n_eval); torch.cuda.synchronize(); edt = time.perf_counter() - t0
        line += f" | eval {n_eval} triples {edt*1e3:.2f} ms -> {n_eval/edt:.0f} test triples/s"
    print(line, flush=True)
    del tr, m
    torch.cuda.empty_cache()
