#!/usr/bin/env python
"""Time the UNMODIFIED reference (read-only at /root/reference) on its CPU-PyTorch path for the bench.py workload
(FB15k-shape TransE d=100 L1, SURVEY.md 8(d) "CPU baseline timing"): BUILD CONTAINER ONLY -- the reference tree cannot
travel to the GPU box, so bench.py times a C/OpenMP port there and quotes this file next to it.

  train: Trainer.train_step_pairwise -> loss.backward() -> optimizer.step() (utils/trainer.py:147-157,298-299) on
         pre-generated batches, B = 32768 positives + 32768 negatives, dense Adam; >= 10 warm-up, median of >= 30
  eval : Evaluator.test loop incl. MetricCalculator (utils/evaluator.py:309-334) on 200 test triples

Writes profiles/r02_reference_cpu_baseline.json.
"""
import json
import os
import platform
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
from pykg2vec.models.pairwise import TransE  # noqa: E402
from pykg2vec.utils.trainer import Trainer  # noqa: E402
from pykg2vec.utils.evaluator import Evaluator  # noqa: E402
from pykg2vec.data.kgcontroller import Triple  # noqa: E402
import bench  # noqa: E402

E, R, DIM, B = bench.E, bench.R, bench.DIM, 32768
train, valid, test = bench.synthetic_split(E, R, (bench.N_TRAIN, bench.N_VALID, bench.N_TEST))
n_eval = 200
q = test[:n_eval]
hr_t, tr_h = bench.build_filters(np.concatenate([train, valid, test]), q, R)
mk = lambda arr: [Triple(int(a), int(b), int(c)) for a, b, c in arr]


class KG:
    def read_cache_data(self, key):
        return {"triplets_test": mk(q), "triplets_valid": mk(valid[:16]), "hr_t": hr_t, "tr_h": tr_h}[key]


cfg = types.SimpleNamespace(tot_entity=E, tot_relation=R, device="cpu", optimizer="adam", learning_rate=0.01, neg_rate=1,
                            alpha=0.1, margin=1.0, batch_size=B, epochs=1000, test_num=n_eval, debug=False,
                            load_from_data=None, hits=[1, 3, 5, 10], patience=3, hidden_size=DIM, l1_flag=True,
                            sampling="uniform", dataset_name="fb15k-shape-synthetic", knowledge_graph=KG())
cfg.summary = lambda: None
torch.manual_seed(0)
model = TransE(**cfg.__dict__)
tr = Trainer(model, cfg)
tr.build_model()
rng = np.random.default_rng(0)
batches = []
for k in range(8):
    pos = train[k * B:(k + 1) * B]
    neg = pos.copy()
    flip = rng.random(B) > 0.5
    rnd = rng.integers(E, size=B)
    neg[:, 2] = np.where(flip, rnd, neg[:, 2])
    neg[:, 0] = np.where(flip, neg[:, 0], rnd)
    batches.append([torch.LongTensor(np.ascontiguousarray(a)) for a in (pos[:, 0], pos[:, 1], pos[:, 2], neg[:, 0], neg[:, 1], neg[:, 2])])


def step(b):
    tr.optimizer.zero_grad()
    loss = tr.train_step_pairwise(*b)
    loss.backward()
    tr.optimizer.step()


model.train()
for k in range(10):
    step(batches[k % 8])
times = []
for k in range(30):
    t0 = time.perf_counter()
    step(batches[k % 8])
    times.append(time.perf_counter() - t0)
med, best = float(np.median(times)), float(np.min(times))
ev = Evaluator(model, cfg)
model.eval()
with torch.no_grad():
    ev.test(ev.test_data, 20, epoch=0)  # warm
    t0 = time.perf_counter()
    ev.test(ev.test_data, n_eval, epoch=0)
    edt = time.perf_counter() - t0
doc = {"what": "unmodified reference (pykg2vec v0.0.52) on torch %s CPU, FB15k-shape TransE d=100 L1, synthetic ids" % torch.__version__,
       "host": "build container: %s, %d logical cores" % (platform.processor() or platform.machine(), os.cpu_count()),
       "cores": torch.get_num_threads(),
       "train": {"value": 2 * B / med, "unit": "scored triples/s", "median_ms_per_step": med * 1e3, "min_ms_per_step": best * 1e3,
                 "sample": "30 timed dense-Adam steps of B=%d positives + %d negatives after 10 warm-up steps (step-only: batches pre-generated)" % (B, B)},
       "eval": {"value": n_eval / edt, "unit": "test triples ranked/s",
                "sample": "Evaluator.test over %d test triples incl. MetricCalculator (two sweeps over E=%d each)" % (n_eval, E)}}
json.dump(doc, open(os.path.join(ROOT, "profiles", "r02_reference_cpu_baseline.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
