"""INTEGRATION.md section 3 end to end, in the build container (the reference tree does not exist on the GPU box):
the reference's OWN Trainer class (pykg2vec/utils/trainer.py: train_model's epoch loop, EarlyStopper, best-metric
save_model, save_training_result, export_embeddings, load_model, tune_model) drives the drop-in models, generator and
evaluator through `pykg2vec_amd.integration.reference_trainer`.  The compute backend is the test-only oracle backend,
so what is under test is the wiring, not the kernels."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


class _KG:
    def __init__(self, cache):
        self.cache = cache
        self.dataset_name = "synthetic"

    def read_cache_data(self, key):
        return self.cache[key]


class Config:  # picklable stand-in for pykg2vec.config.Config (np.save(config.npy, config), utils/trainer.py:396)
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def summary(self):
        pass


def _config(tmp, c, model_name, **extra):
    hr_t, tr_h = c.filters()
    for sub in ("tmp", "result", "emb"):
        (tmp / sub).mkdir(exist_ok=True)
    kw = dict(tot_entity=c.E, tot_relation=c.R, device="cpu", optimizer="adam", learning_rate=0.05, neg_rate=1, alpha=0.1,
              margin=1.0, batch_size=64, epochs=4, test_num=8, test_step=1, debug=False, hits=[1, 3, 5, 10], patience=1,
              dataset_name="synthetic", sampling="uniform", tot_train_triples=len(c.train), seed=0, load_from_data=None,
              save_model=True, disp_result=False, model_name=model_name, path_tmp=tmp / "tmp", path_result=tmp / "result",
              path_embeddings=tmp / "emb",
              knowledge_graph=_KG({"triplets_train": c.train, "triplets_valid": c.valid, "triplets_test": c.test,
                                   "hr_t": hr_t, "tr_h": tr_h,
                                   "idx2entity": {i: "e%d" % i for i in range(c.E)},
                                   "idx2relation": {i: "r%d" % i for i in range(c.R)}}))
    kw.update(c.hp)
    kw.update(extra)
    return Config(**kw)


@pytest.fixture
def installed_models():
    """The drop-in classes patched into the reference's model modules for ONE test: the reference's own classes are put back
    whatever the test does (round 5 left them patched: tests importing the reference after this file then built HIP models)."""
    ref_shim.install()
    from pykg2vec_amd import integration
    import pykg2vec.models.pairwise as ref_pw
    original = ref_pw.TransE
    handle = integration.install_models()
    try:
        yield handle
    finally:
        handle.restore()
        assert ref_pw.TransE is original and ref_pw.TransE.__module__ == "pykg2vec.models.pairwise"


def test_install_models_is_undone_by_its_handle():
    ref_shim.install()
    from pykg2vec_amd import integration
    import pykg2vec.models.pairwise as ref_pw
    import pykg2vec.models.pointwise as ref_pt
    before = {(m, n): getattr(m, n) for m, names in ((ref_pw, integration.PAIRWISE), (ref_pt, integration.POINTWISE)) for n in names}
    with integration.install_models():
        assert ref_pw.TransE.__module__ == "pykg2vec_amd.pairwise" and ref_pt.Complex.__module__ == "pykg2vec_amd.pointwise"
        with integration.install_models():          # nested / repeated installs keep the ORIGINAL classes on record
            pass
        assert ref_pw.TransE.__module__ == "pykg2vec_amd.pairwise"
    for (m, n), cls in before.items():
        assert getattr(m, n) is cls, n


@pytest.mark.parametrize("case,model_name", [("transe_l1", "TransE"), ("distmult", "DistMult")])
def test_reference_trainer_drives_the_drop_in_path(tmp_path, case, model_name, installed_models):
    ref_shim.install()
    import oracle_backend
    from golden_util import Case
    from pykg2vec.common import Importer
    import pykg2vec.utils.trainer as ref_tr
    from pykg2vec_amd import integration
    import pykg2vec_amd.evaluator as hip_ev
    import pykg2vec_amd.generator as hip_gen

    c = Case(case)
    cfg = _config(tmp_path, c, model_name)
    _, model_def = Importer().import_model_config(model_name.lower())   # the reference's own discovery finds OUR class
    assert model_def.__module__.startswith("pykg2vec_amd.")
    torch.manual_seed(0)
    model = model_def(**cfg.__dict__)
    Trainer = integration.reference_trainer(backend=oracle_backend)
    assert issubclass(Trainer, ref_tr.Trainer)
    tr = Trainer(model, cfg)
    tr.build_model()
    assert isinstance(tr.early_stopper, ref_tr.EarlyStopper) and isinstance(tr.evaluator, hip_ev.Evaluator)
    before = model.ent_embeddings.weight.detach().clone()
    last = tr.train_model()                       # the reference's loop: epochs, mini_test, early stop, save, export
    assert 0 <= last < cfg.epochs
    assert isinstance(tr.generator, hip_gen.Generator)
    assert ref_tr.Generator.__module__ == "pykg2vec.data.generator"      # module globals restored
    assert not torch.equal(before, model.ent_embeddings.weight)           # the fused loop trained the tables
    losses = [l for _, l in tr.training_results]
    assert len(losses) == last + 1 and losses[-1] < losses[0]
    # artefacts written by the reference's own persistence code over the drop-in model
    saved = cfg.path_tmp / model.model_name
    assert (saved / "model.vec.pt").exists() and (saved / "config.npy").exists()
    assert any("Training_results" in f for f in os.listdir(cfg.path_result))
    assert (cfg.path_embeddings / model.model_name / ("%s.tsv" % model.parameter_list[0].name)).exists()
    assert tr.evaluator.metric_calculator.fmr  # metric dict fields filled by the drop-in evaluator

    # reference load_model: rebuilds the model through Importer from the pickled config and loads the state dict
    cfg2 = _config(tmp_path, c, model_name, load_from_data=str(saved))
    tr2 = Trainer(model_def(**cfg2.__dict__), cfg2)
    tr2.build_model()
    sd = torch.load(str(saved / "model.vec.pt"))
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert tr2.flat.param.numel() >= sum(v.numel() for v in sd.values())   # re-homed into the flat buffer after loading

    # reference tune_model over the same graft
    cfg3 = _config(tmp_path, c, model_name, epochs=1)
    tr3 = Trainer(model_def(**cfg3.__dict__), cfg3)
    tr3.build_model()
    assert np.isfinite(tr3.tune_model())
