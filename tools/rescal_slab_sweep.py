"""Dev tool: the pairwise RESCAL step below the large-batch thresholds, tile kernel (k_rescal_pair, KGE_RESCAL_SLAB=0) against the slab
form (kge_rescal_slab.hip), same process and box, over batch sizes / widths / relation counts.  us per step through
Trainer.train_model_epoch (whole step incl. sampler, grouping, optimiser)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_util
from pykg2vec_amd.trainer import Trainer

SHAPES = {"yago310": (123182, 37, 1079040), "fb15k": (14951, 1345, 483142), "wn18rr": (40943, 11, 86835)}
CASES = [("yago310", 200, 128), ("yago310", 200, 512), ("yago310", 200, 1024), ("yago310", 200, 2048), ("yago310", 200, 4096),
         ("yago310", 100, 1024), ("yago310", 50, 1024), ("yago310", 256, 1024), ("wn18rr", 200, 1024), ("wn18rr", 50, 128),
         ("fb15k", 50, 128), ("fb15k", 200, 256), ("fb15k", 100, 448)]
rng = np.random.default_rng(7)
for ds, k, B in CASES:
    E, R, NTR = SHAPES[ds]
    train = np.stack([rng.integers(E, size=NTR), rng.integers(R, size=NTR), rng.integers(E, size=NTR)], 1)
    test = train[:16]
    out = {}
    for arm in ("0", "1", "0", "1"):
        os.environ["KGE_RESCAL_SLAB"] = arm
        hp = dict(hidden_size=k, margin=1.0)
        cfg = hip_util.make_config(E, R, dict(hp, neg_rate=1), train, test, test, optimizer="adam", lr=0.01, batch_size=B)
        torch.manual_seed(0)
        m = hip_util.model_from_params("rescal", {}, hp, E, R, train=train)
        tr = Trainer(m, cfg)
        tr.build_model()
        tr.generator = tr._new_generator()
        K_ = min(200, max(1, NTR // B))
        cfg.tot_train_triples = B * K_
        tr.train_model_epoch(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_model_epoch(1)
        torch.cuda.synchronize()
        out.setdefault(arm, []).append((time.perf_counter() - t0) / K_ * 1e6)
        del tr, m
        torch.cuda.empty_cache()
    print("%-8s k=%3d B=%5d  tile kernel %s us | slab form %s us" % (ds, k, B, " / ".join("%.1f" % v for v in out["0"]),
                                                                      " / ".join("%.1f" % v for v in out["1"])), flush=True)
