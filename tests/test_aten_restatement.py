"""oracle/aten_step.py (the ATen-op-for-op restatement bench.py times as `cpu_baseline.kind = "aten-restatement"` where the reference
tree is absent) must be BIT-EQUAL to the live reference: same loss, same gradients, same weights after several optimiser steps, same
raw and filtered ranks.  Live comparison in the build container; the frozen fixture (tests/golden/ref_aten_step.npz, written from the
live reference by oracle/make_golden_aten.py) keeps the check alive where the tree is absent."""
import os
import types

import numpy as np
import pytest
import torch

import aten_step
import ref_shim

E, R, D, B, N_TRAIN, N_TEST = 300, 7, 20, 64, 640, 16


def _graph(seed=3):
    rng = np.random.default_rng(seed)
    draw = lambda n: np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1).astype(np.int64)
    train, test = draw(N_TRAIN), draw(N_TEST)
    hr_t, tr_h = {}, {}
    for h, r, t in np.concatenate([train, test]):
        hr_t.setdefault((int(h), int(r)), set()).add(int(t))
        tr_h.setdefault((int(t), int(r)), set()).add(int(h))
    return train, test, hr_t, tr_h


def run_restatement(l1, opt_name, steps=3):
    """Deterministic run of the restatement: returns dict of arrays (losses, grads of step 0, weights after `steps`, ranks)."""
    train, test, hr_t, tr_h = _graph()
    torch.manual_seed(11)
    m = aten_step.AtenTransE(E, R, D, l1)
    opt = aten_step.make_optimizer(m, opt_name, 0.05)
    batches = aten_step.corrupt_batches(train, E, B, steps, seed=5)
    out = {"init_ent": m.ent_embeddings.weight.detach().numpy().copy()}
    losses = []
    for k, b in enumerate(batches):
        losses.append(aten_step.train_step(m, opt, b, 1.0).item())
        if k == 0:
            out["grad_ent"] = m.ent_embeddings.weight.grad.numpy().copy()
            out["grad_rel"] = m.rel_embeddings.weight.grad.numpy().copy()
    out["losses"] = np.array(losses, dtype=np.float32)
    out["ent"] = m.ent_embeddings.weight.detach().numpy().copy()
    out["rel"] = m.rel_embeddings.weight.detach().numpy().copy()
    out["ranks"] = aten_step.rank_pass(m, test, hr_t, tr_h, E)
    return out


def run_reference(l1, opt_name, steps=3):
    """The same run through the UNMODIFIED reference classes (container only)."""
    ref_shim.install()
    from pykg2vec.models.pairwise import TransE
    from pykg2vec.utils.trainer import Trainer
    from pykg2vec.utils.evaluator import Evaluator
    from pykg2vec.data.kgcontroller import Triple
    import contextlib
    import io
    train, test, hr_t, tr_h = _graph()
    cache = {"triplets_test": [Triple(int(a), int(b), int(c)) for a, b, c in test], "triplets_valid": [], "hr_t": hr_t, "tr_h": tr_h}
    kg = types.SimpleNamespace(read_cache_data=lambda key: cache[key])
    cfg = types.SimpleNamespace(tot_entity=E, tot_relation=R, device="cpu", optimizer=opt_name, learning_rate=0.05, neg_rate=1,
                                alpha=0.1, margin=1.0, batch_size=B, epochs=1000, test_num=N_TEST, debug=False, load_from_data=None,
                                hits=[1, 3, 5, 10], patience=3, hidden_size=D, l1_flag=l1, sampling="uniform",
                                dataset_name="synthetic", knowledge_graph=kg)
    cfg.summary = lambda: None
    torch.manual_seed(11)
    model = TransE(**cfg.__dict__)
    tr = Trainer(model, cfg)
    tr.build_model()
    batches = aten_step.corrupt_batches(train, E, B, steps, seed=5)
    out = {"init_ent": model.ent_embeddings.weight.detach().numpy().copy()}
    losses = []
    for k, b in enumerate(batches):   # utils/trainer.py:266-299
        model.train()
        tr.optimizer.zero_grad()
        loss = tr.train_step_pairwise(*b)
        loss.backward()
        tr.optimizer.step()
        losses.append(loss.item())
        if k == 0:
            out["grad_ent"] = model.ent_embeddings.weight.grad.numpy().copy()
            out["grad_rel"] = model.rel_embeddings.weight.grad.numpy().copy()
    out["losses"] = np.array(losses, dtype=np.float32)
    out["ent"] = model.ent_embeddings.weight.detach().numpy().copy()
    out["rel"] = model.rel_embeddings.weight.detach().numpy().copy()
    ev = Evaluator(model, cfg)
    model.eval()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ev.test(ev.test_data, N_TEST, epoch=0)
    mc = ev.metric_calculator
    out["ranks"] = np.array([mc.rank_head, mc.rank_tail, mc.f_rank_head, mc.f_rank_tail], dtype=np.int64)
    return out


CASES = [(True, "adam"), (False, "adam"), (True, "sgd"), (True, "adagrad"), (False, "rms")]


@pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")
@pytest.mark.parametrize("l1,opt_name", CASES)
def test_bit_equal_to_live_reference(l1, opt_name):
    got, ref = run_restatement(l1, opt_name), run_reference(l1, opt_name)
    for k in ref:
        assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape, k
        assert np.array_equal(got[k], ref[k]), (k, np.abs(got[k].astype(np.float64) - ref[k]).max())


@pytest.mark.parametrize("l1,opt_name", CASES)
def test_bit_equal_to_frozen_reference_outputs(golden_dir, l1, opt_name):
    z = np.load(os.path.join(golden_dir, "ref_aten_step.npz"))
    if str(z["torch_version"]) != torch.__version__:
        pytest.skip("fixture frozen on torch %s, running %s: bit equality is only promised on the same build" % (z["torch_version"], torch.__version__))
    got = run_restatement(l1, opt_name)
    tag = "%s_%s_" % ("l1" if l1 else "l2", opt_name)
    for k, v in got.items():
        assert np.array_equal(v, z[tag + k]), (tag + k)
