"""GPU-side test helpers: build drop-in models from golden cases / oracle parameter dicts."""
import types

import numpy as np
import torch

import pykg2vec_amd as pa
from golden_util import Case

CLASS_OF = {"transe": "transe", "transh": "transh", "transd": "transd", "rotate": "rotate", "rescal": "rescal",
            "ntn": "ntn", "distmult": "distmult", "complex": "complex", "complexn3": "complexn3", "analogy": "analogy"}
DEV = "cuda"


class KG:
    def __init__(self, cache):
        self.cache = cache
        self.dataset_name = "synthetic"

    def read_cache_data(self, key):
        return self.cache[key]


def make_config(E, R, hp, train, valid, test, optimizer="sgd", lr=0.05, batch_size=32, device=DEV, **extra):
    allt = np.concatenate([train, valid, test])
    hr_t, tr_h = {}, {}
    for h, r, t in allt:
        hr_t.setdefault((int(h), int(r)), set()).add(int(t))
        tr_h.setdefault((int(t), int(r)), set()).add(int(h))
    cfg = types.SimpleNamespace(
        tot_entity=E, tot_relation=R, device=device, optimizer=optimizer, learning_rate=lr,
        neg_rate=hp.get("neg_rate", 1), alpha=hp.get("alpha", 0.1), margin=hp.get("margin", 1.0),
        batch_size=batch_size, epochs=1000, test_num=0, test_step=1, debug=False, hits=[1, 3, 5, 10], patience=3,
        dataset_name="synthetic", sampling="uniform", tot_train_triples=len(train), seed=0,
        knowledge_graph=KG({"triplets_train": train, "triplets_valid": valid, "triplets_test": test,
                            "hr_t": hr_t, "tr_h": tr_h}))
    for k, v in hp.items():
        setattr(cfg, k, v)
    for k, v in extra.items():
        setattr(cfg, k, v)
    return cfg


def model_from_params(model_name, params, hp, E, R, device=DEV, train=None):
    cls = pa.import_model(model_name)
    kw = dict(hp)
    kw.update(tot_entity=E, tot_relation=R)
    # constructor inputs some models read besides their hyper-parameters (TransM: train split + device,
    # pairwise.py:299-315; SimplE: pointwise.py:478-479)
    kw.setdefault("device", device)
    kw.setdefault("batch_size", 32)
    kw.setdefault("tot_train_triples", 0 if train is None else len(train))
    if train is not None:
        kw.setdefault("knowledge_graph", KG({"triplets_train": train}))
    m = cls(**kw)
    with torch.no_grad():
        for k, v in params.items():
            getattr(m, k).weight.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return m.to(device)


def model_from_case(c, prefix="init.", device=DEV):
    return model_from_params(c.model, c.params(prefix), c.hp, c.E, c.R, device, train=c.train)


def table_parameters(model):
    """(state_dict name, parameter) of the model's embedding tables, parameter_list order (QuatE also registers
    fc / bn modules its forward never uses)."""
    named = {id(q): n for n, q in model.named_parameters()}
    return [(named[id(p.weight)], p.weight) for p in model.parameter_list]


def dev(a, dtype=torch.int64):
    return torch.as_tensor(np.asarray(a), dtype=dtype, device=DEV)


def params_of(model):
    return {p_name: getattr(model, p_name).weight.detach().cpu().numpy()
            for p_name in [n.split(".")[0] for n, _ in model.named_parameters()]}


def record_max(report, key, value):
    """Keep the largest `value` seen under `key` in gpurun_out/<report>.json (observed deviations from the reference, copied into
    profiles/ by the builder: the tolerances of the deterministic paths are set to ~2x what is recorded there)."""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    fn = os.path.join(out, report + ".json")
    doc = json.load(open(fn)) if os.path.exists(fn) else {}
    doc[key] = max(float(value), doc.get(key, 0.0))
    json.dump(doc, open(fn, "w"), indent=1, sort_keys=True)


def train_fullsize(name):
    """golden_util.TRAINED[name]: the seeded full-size tables after `epochs` epochs of the drop-in Trainer's default step path over
    the whole synthetic train split -> (tables {name: float32 array}, model, spec, (train, valid, test)).  Bit-reproducible: the
    owner-computes / two-phase own / staged steps involve no float atomics and the generator's permutation and Philox counters are
    functions of (seed, epoch, batch)."""
    import golden_util as gu
    from pykg2vec_amd.trainer import Trainer
    spec, P, train, valid, test, _ids, _batch = gu.fullsize_inputs(name)
    t = gu.TRAINED[name]
    hp = dict(spec["hp"])
    hp.setdefault("margin", 1.0)
    cfg = make_config(spec["E"], spec["R"], hp, train[:1], valid[:4], test[:4], optimizer=t["optimizer"], lr=t["lr"], batch_size=t["B"])
    cfg.knowledge_graph.cache.update(triplets_train=train)
    cfg.seed, cfg.tot_train_triples, cfg.sampling, cfg.epochs = gu.GENERATOR_SEED, len(train), "uniform", t["epochs"]
    m = model_from_params(spec["model"], P, spec["hp"], spec["E"], spec["R"], train=train)
    tr = Trainer(m, cfg)
    tr.build_model()
    tr.generator = tr._new_generator()
    losses = [float(tr.train_model_epoch(e)) for e in range(t["epochs"])]
    tr.sync_model()
    torch.cuda.synchronize()
    path = ("hipGraph" if tr._graph is not None else "staged" if getattr(tr, "_staged", None) is not None else
            "own" if getattr(tr, "_own", None) is not None else "pull" if getattr(tr, "_pull", None) is not None else "eager")
    tables = {k[:-len(".weight")]: p.detach().cpu().numpy().copy() for k, p in table_parameters(m)}
    return tables, m, spec, (train, valid, test), dict(path=path, losses=losses)
