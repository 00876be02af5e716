"""Time the UNMODIFIED reference (pykg2vec at $PYKG2VEC_REFERENCE, default /root/reference) on its CPU-PyTorch path for the
bench.py workload -- TEST / MEASUREMENT INFRASTRUCTURE, never part of the product path.  SURVEY.md section 8(d), "CPU baseline
timing":

  train: Trainer.train_step_pairwise -> loss.backward() -> optimizer.step() (utils/trainer.py:147-157,298-299) on
         pre-generated batches of B positives + B negatives, dense Adam; >= 10 warm-up steps, median of the timed ones
  eval : Evaluator.test loop incl. MetricCalculator (utils/evaluator.py:309-334) on n_eval test triples

`measure()` is called by bench.py's cpu_baseline leg when the reference tree can be imported (`available()`; it cannot travel
to the GPU box, where bench.py falls back to the C/OpenMP port and says so) and by tools/ref_cpu_baseline.py, which writes
profiles/r04_reference_cpu_baseline.json in the build container.
"""
import os
import platform
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import ref_shim  # noqa: E402


def available():
    return ref_shim.reference_available()


def measure(E, R, dim, train, valid, test, hr_t, tr_h, batch=32768, n_eval=200, train_budget_s=12.0, min_timed=5, max_timed=30):
    """Returns {"train": {...}, "eval": {...}, "cores", "host", "what"}.  hr_t / tr_h: the filter dicts for test[:n_eval]."""
    ref_shim.install()
    import torch
    from pykg2vec.models.pairwise import TransE
    from pykg2vec.utils.trainer import Trainer
    from pykg2vec.utils.evaluator import Evaluator
    from pykg2vec.data.kgcontroller import Triple

    q = test[:n_eval]
    mk = lambda arr: [Triple(int(a), int(b), int(c)) for a, b, c in arr]
    cache = {"triplets_test": mk(q), "triplets_valid": mk(valid[:16]), "hr_t": hr_t, "tr_h": tr_h}

    class KG:
        def read_cache_data(self, key):
            return cache[key]

    cfg = types.SimpleNamespace(tot_entity=E, tot_relation=R, device="cpu", optimizer="adam", learning_rate=0.01, neg_rate=1,
                                alpha=0.1, margin=1.0, batch_size=batch, epochs=1000, test_num=n_eval, debug=False,
                                load_from_data=None, hits=[1, 3, 5, 10], patience=3, hidden_size=dim, l1_flag=True,
                                sampling="uniform", dataset_name="fb15k-shape-synthetic", knowledge_graph=KG())
    cfg.summary = lambda: None
    torch.manual_seed(0)
    model = TransE(**cfg.__dict__)
    tr = Trainer(model, cfg)
    tr.build_model()
    rng = np.random.default_rng(0)
    batches = []
    for k in range(min(8, len(train) // batch)):
        pos = train[k * batch:(k + 1) * batch]
        neg = pos.copy()
        flip = rng.random(batch) > 0.5
        rnd = rng.integers(E, size=batch)
        neg[:, 2] = np.where(flip, rnd, neg[:, 2])
        neg[:, 0] = np.where(flip, neg[:, 0], rnd)
        batches.append([torch.LongTensor(np.ascontiguousarray(a)) for a in (pos[:, 0], pos[:, 1], pos[:, 2], neg[:, 0], neg[:, 1], neg[:, 2])])

    def step(b):   # the body of the reference's epoch loop (utils/trainer.py:296-299)
        tr.optimizer.zero_grad()
        loss = tr.train_step_pairwise(*b)
        loss.backward()
        tr.optimizer.step()

    model.train()
    for k in range(10):
        step(batches[k % len(batches)])
    times, t_begin = [], time.perf_counter()
    while len(times) < max_timed and (len(times) < min_timed or time.perf_counter() - t_begin < train_budget_s):
        t0 = time.perf_counter()
        step(batches[len(times) % len(batches)])
        times.append(time.perf_counter() - t0)
    med, best = float(np.median(times)), float(np.min(times))
    ev = Evaluator(model, cfg)
    model.eval()
    import contextlib
    import io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ev.test(ev.test_data, min(20, n_eval), epoch=0)  # warm
        t0 = time.perf_counter()
        ev.test(ev.test_data, n_eval, epoch=0)
        edt = time.perf_counter() - t0
    return {"what": "unmodified reference (pykg2vec at %s) on torch %s CPU, FB15k-shape TransE d=%d L1, synthetic ids"
                    % (ref_shim.REFERENCE_ROOT, torch.__version__, dim),
            "host": "%s, %d logical cores" % (platform.processor() or platform.machine(), os.cpu_count()),
            "cores": torch.get_num_threads(),
            "train": {"value": 2 * batch / med, "unit": "scored triples/s", "median_ms_per_step": med * 1e3, "min_ms_per_step": best * 1e3,
                      "sample": "%d timed dense-Adam steps of B=%d positives + %d negatives after 10 warm-up steps, median (step-only: "
                                "Trainer.train_step_pairwise + backward + optimizer.step on pre-generated batches)" % (len(times), batch, batch)},
            "eval": {"value": n_eval / edt, "unit": "test triples ranked/s",
                     "sample": "Evaluator.test over %d test triples incl. MetricCalculator (two sweeps over E=%d each)" % (n_eval, E)}}
