#!/usr/bin/env python
"""Learning curves of the LIVE reference on a planted graph (container-only; TEST INFRASTRUCTURE) -> tests/golden/ref_trajectory.npz.

Per-step parity is pinned elsewhere; the sampler's RNG differs from the reference's by design, so what multi-epoch training does
can only be compared as a DISTRIBUTION.  This script runs the reference's own Trainer (utils/trainer.py:190-239: Generator with its
worker processes, train_model_epoch, torch.optim) for `epochs` epochs x TRAJECTORY_SEEDS seeds on tests/golden_util.planted_graph
for each configuration of golden_util.TRAJECTORY (TransE L1 / Adam, ComplEx / Adagrad, RotatE / Adam with self-adversarial
negatives), evaluates the held-out test triples with the reference's Evaluator.test after EVERY epoch, and freezes per epoch and seed
the epoch loss and the filtered MR / MRR / Hits@10.  tests/test_hip_trajectory.py trains the drop-in Trainer on the same graph and
must land inside mean +- 3 sigma (plus a stated floor) of these curves.

Usage: python oracle/make_golden_trajectory.py [config ...]"""
import contextlib
import io
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
from pykg2vec.utils.trainer import Trainer  # noqa: E402
from pykg2vec.data.generator import Generator  # noqa: E402
from pykg2vec.data.kgcontroller import Triple  # noqa: E402
import golden_util as gu  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_trajectory.npz")


class _KG:
    def __init__(self, cache):
        self.cache = cache
        self.dataset_name = "planted"

    def read_cache_data(self, key):
        return self.cache[key]


class Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def summary(self):
        pass


def run(name, seed, graph, tmp):
    c = gu.TRAJECTORY[name]
    E, R, train, valid, test = graph
    mk = lambda arr: [Triple(int(a), int(b), int(c_)) for a, b, c_ in arr]
    hr_t, tr_h = {}, {}
    for h, r, t in np.concatenate([train, valid, test]):
        hr_t.setdefault((int(h), int(r)), set()).add(int(t))
        tr_h.setdefault((int(t), int(r)), set()).add(int(h))
    for sub in ("tmp", "result", "emb"):
        (tmp / sub).mkdir(exist_ok=True)
    cfg = Config(tot_entity=E, tot_relation=R, device="cpu", optimizer=c["optimizer"], learning_rate=c["lr"], neg_rate=c["neg"],
                 alpha=c["hp"].get("alpha", 0.1), margin=c["hp"].get("margin", 1.0), batch_size=c["batch"], epochs=10 ** 6, test_num=len(test),
                 test_step=1, debug=False, hits=[1, 3, 5, 10], patience=10 ** 6, dataset_name="planted", sampling="uniform",
                 tot_train_triples=len(train), load_from_data=None, save_model=False, disp_result=False, num_process_gen=2,
                 model_name=c["model"], path_tmp=tmp / "tmp", path_result=tmp / "result", path_embeddings=tmp / "emb",
                 knowledge_graph=_KG({"triplets_train": mk(train), "triplets_valid": mk(valid), "triplets_test": mk(test), "hr_t": hr_t,
                                      "tr_h": tr_h, "relationproperty": {r: 0.5 for r in range(R)}}))
    for k, v in c["hp"].items():
        setattr(cfg, k, v)
    torch.manual_seed(seed)
    np.random.seed(seed)              # (the generator's worker processes fork this state)
    mod, cls = c["ref"].split(".")
    model = getattr(__import__("pykg2vec.models." + mod, fromlist=[cls]), cls)(**cfg.__dict__)
    tr = Trainer(model, cfg)
    tr.build_model()
    tr.generator = Generator(model, cfg)       # utils/trainer.py:190
    out = {k: [] for k in ("loss", "fmr", "fmrr", "fhit10", "mr")}
    sink = io.StringIO()
    try:
        for e in range(c["epochs"]):
            with contextlib.redirect_stderr(sink), contextlib.redirect_stdout(sink):
                loss = tr.train_model_epoch(e)
                model.eval()
                with torch.no_grad():
                    tr.evaluator.test(tr.evaluator.test_data, len(test), epoch=e)
            mc = tr.evaluator.metric_calculator
            out["loss"].append(float(loss)); out["fmr"].append(float(mc.fmr[e])); out["fmrr"].append(float(mc.fmrr[e]))
            out["fhit10"].append(float(mc.fhit[(e, 10)])); out["mr"].append(float(mc.mr[e]))
    finally:
        tr.generator.stop()
    return {k: np.asarray(v, np.float64) for k, v in out.items()}


if __name__ == "__main__":
    names = sys.argv[1:] or list(gu.TRAJECTORY)
    graph = gu.planted_graph()
    doc = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    with tempfile.TemporaryDirectory() as d:
        for name in names:
            runs = [run(name, s, graph, Path(d)) for s in range(gu.TRAJECTORY_SEEDS)]
            for k in runs[0]:
                doc["%s.%s" % (name, k)] = np.stack([r[k] for r in runs])          # [seeds, epochs]
            last = {k: doc["%s.%s" % (name, k)][:, -1] for k in ("fmr", "fmrr", "fhit10")}
            print(name, "final epoch over seeds:", {k: (round(float(v.mean()), 4), round(float(v.std()), 4)) for k, v in last.items()}, flush=True)
    np.savez_compressed(OUT, **doc)
    print("wrote", OUT)
