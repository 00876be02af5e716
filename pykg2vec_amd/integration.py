"""Drop-in wiring into an importable reference tree, without editing it.

`install_models()` puts this package's model classes where `Importer().import_model_config` looks them up
(pykg2vec/common.py:300-328 -> `pykg2vec.models.pairwise` / `.pointwise`).  `reference_trainer()` returns a subclass
of the reference's own `pykg2vec.utils.trainer.Trainer` whose hot loop (`build_model`, `train_model_epoch`, the
`train_step_*` methods and their private helpers) is this package's fused path, while everything outside the hot
path -- `train_model`'s epoch / early-stopping / best-checkpoint logic, `tune_model`, `save_model`, `load_model`,
`infer_*`, `export_embeddings`, `save_training_result`, `display` -- stays the reference's code, running over the
drop-in models, generator and evaluator.
"""
import types

PAIRWISE = ("TransE", "TransH", "TransD", "TransM", "TransR", "RotatE", "Rescal", "NTN")
POINTWISE = ("DistMult", "Complex", "ComplexN3", "ANALOGY", "CP", "SimplE", "SimplE_ignr", "QuatE")


class _Installed:
    """Undo handle of `install_models()`: `.restore()` puts the reference's own classes back (idempotent); also a context
    manager (`with install_models(): ...`)."""

    def __init__(self, saved):
        self._saved = saved

    def restore(self):
        for module, name, cls in self._saved:
            setattr(module, name, cls)
        self._saved = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.restore()
        return False


def install_models():
    """Patch the drop-in classes into `pykg2vec.models.pairwise / .pointwise`; returns the undo handle.  A process that only
    ever wants the HIP models can ignore it; anything that also uses the reference's own classes (the parity tests) restores."""
    import pykg2vec.models.pairwise as ref_pw
    import pykg2vec.models.pointwise as ref_pt
    from . import pairwise as pw, pointwise as pt
    saved = []
    for ref_mod, mod, names in ((ref_pw, pw, PAIRWISE), (ref_pt, pt, POINTWISE)):
        for name in names:
            ours = getattr(mod, name)
            if getattr(ref_mod, name) is not ours:     # (a second install must not record our own class as "the original")
                saved.append((ref_mod, name, getattr(ref_mod, name)))
            setattr(ref_mod, name, ours)
    return _Installed(saved)


def reference_trainer(backend=None, process_group=None, use_graph=None):
    """The reference Trainer class with the MI355X hot loop grafted in.  `backend` / `process_group` / `use_graph` are
    forwarded to the hot path (see pykg2vec_amd.trainer.Trainer)."""
    import pykg2vec.utils.trainer as ref_tr
    from .common import Monitor
    from .evaluator import Evaluator as HipEvaluator
    from .generator import Generator as HipGenerator
    from .trainer import Trainer as Hip

    class Trainer(ref_tr.Trainer):
        GRAPH_MAX_ROWS, GRAPH_UNROLL = Hip.GRAPH_MAX_ROWS, Hip.GRAPH_UNROLL

        def __init__(self, model, config):
            super().__init__(model, config)
            self._init_hot_path(process_group, backend, use_graph)

        def build_model(self, monitor=Monitor.FILTERED_MEAN_RANK):
            if getattr(self.config, "load_from_data", None) is not None:   # utils/trainer.py:105-106
                self.load_model(self.config.load_from_data)
            Hip.build_model(self, monitor)
            # the reference's own stopper, keyed by the reference's own enum (this package defines stand-in enums when
            # it was imported before the reference became importable)
            self.early_stopper = ref_tr.EarlyStopper(self.config.patience, ref_tr.Monitor(monitor.value))
            if hasattr(self.config, "summary"):
                self.config.summary()

        # the reference's train_model / tune_model construct `Generator(model, config)` and `Evaluator(model, config,
        # tuning=True)` from their module globals (utils/trainer.py:191,245-246): swap in the device-resident ones for
        # the duration of the call
        def _with_hip_collaborators(self, fn):
            saved = ref_tr.Generator, ref_tr.Evaluator
            ref_tr.Generator = lambda model, config: HipGenerator(model, config, rank=self.rank, world_size=self.world_size,
                                                                  backend=self.K)
            ref_tr.Evaluator = lambda model, config, tuning=False: HipEvaluator(model, config, tuning=tuning, backend=self.K)
            try:
                return fn()
            finally:
                ref_tr.Generator, ref_tr.Evaluator = saved

        def train_model(self):
            return self._with_hip_collaborators(lambda: ref_tr.Trainer.train_model(self))

        def tune_model(self):
            return self._with_hip_collaborators(lambda: ref_tr.Trainer.tune_model(self))

    # the hot loop, its private helpers and the constants they read, taken from the class that owns them (no name list of the
    # private part to keep in sync); the reference keeps everything outside the hot path
    public = ("train_model_epoch", "train_step_pairwise", "train_step_pointwise", "step_next_batch", "step_next_batches", "step_path",
              "pull_step_explicit", "own_step_explicit", "transx_step_explicit", "sync_model")
    for name, obj in vars(Hip).items():
        if name in vars(Trainer) or name.startswith("__"):
            continue
        if isinstance(obj, types.FunctionType):
            if name.startswith("_") or name in public:
                setattr(Trainer, name, obj)
        elif name.isupper() and name not in ("TRAINED_MODEL_FILE_NAME", "TRAINED_MODEL_CONFIG_NAME"):
            setattr(Trainer, name, obj)        # STEP_PATHS, PULL_INDEX_BUDGET, OWN_GENERIC_MODELS, ...
    return Trainer
