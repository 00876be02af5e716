"""Drop-in Trainer for the hot loop of pykg2vec/utils/trainer.py (build_model :103-144, train_step_* :147-180,
train_model_epoch :259-307, train_model :182-239, tune_model :241-257) on the fused MI355X path.

Per step the reference issues ~30 ATen kernels, two autograd graphs, a dense torch.optim sweep per table and a
`loss.item()` sync.  Here a step is: one fused score+loss+backward kernel (gradients scattered into ONE flat dense
buffer), at most one all-reduce of that buffer (data parallel over RCCL), one fused dense optimiser sweep over ONE
flat parameter buffer (which also clears the gradients), and no host synchronisation until the epoch ends.

Inference helpers, hyper-parameter tuning, early stopping, export and plotting are outside the hot path (SURVEY.md
section 2) and stay with the reference: `pykg2vec_amd.integration.reference_trainer()` grafts this module's hot loop onto
the reference's own Trainer class, which keeps all of those (tests/test_integration_graft.py).  The stand-alone Trainer
keeps what `train_model` itself needs to honour the configuration: the patience rule of the reference's early stopping
(`PatienceStopper`; any object with `should_stop(metrics)` can be injected -- the graft passes the reference's own
EarlyStopper), best-metric checkpointing under `config.save_model`, and the checkpoint pair in the reference's format
(save_model / load_model: `model.vec.pt` state_dict + `config.npy`).
"""
import os

import torch

from . import kernels as K
from .common import Monitor, TrainingStrategy
from .evaluator import Evaluator
from .generator import Generator


from .state import FlatState, PatienceStopper, PullState, _log   # noqa: F401  (re-exported: tests and the graft import them from here)


class Trainer:
    TRAINED_MODEL_FILE_NAME = "model.vec.pt"      # utils/trainer.py:86-87
    TRAINED_MODEL_CONFIG_NAME = "config.npy"

    GRAPH_MAX_ROWS = 16384  # steps scoring at most this many triples are launch-bound: replay them as one hipGraph
    # bytes the per-batch incidence index of an owner-computes path (+ its sampler list sets, + the build's workspace) may occupy:
    # at most this, and at most a quarter of the device memory free when the path is chosen (see _owner_index_fits).  Beyond it
    # the step falls back to the atomic-scatter kernels (hipGraph-replayed when launch-bound), which need no index.
    PULL_INDEX_BUDGET = 32 << 30
    GRAPH_UNROLL = 8        # steps per replayed multi-step graph (even; 0 = single-step graphs only)
    OWN_GENERIC_MODELS = ("analogy", "cp", "simple", "simple_ignr", "quate")
    PULL_TWO_PHASE_MIN_BATCH = 8192    # owner-computes step in two launches (each pair evaluated once) from this batch size on

    def __init__(self, model, config, process_group=None, backend=None, use_graph=None):
        self.model = model
        self.config = config
        self.training_results = []
        self.evaluator = None
        self.generator = None
        self.early_stopper = None
        self.monitor = None
        self._init_hot_path(process_group, backend, use_graph)

    def _switches(self):
        """The A/B switches of DESIGN.md section 5a, read ONCE when a Trainer is constructed (None = unset: the built-in
        rule decides).  They exist for same-box A/B runs and for tests that force a path; the hot loop never reads the
        environment."""
        env = os.environ.get
        flag = lambda name: None if env(name) is None else env(name) == "1"
        return {"pull": flag("KGE_PULL"), "staged": flag("KGE_STAGED"), "graph_multi": flag("KGE_GRAPH_MULTI"),
                "pw_pull": flag("KGE_PW_PULL"), "rescal_fused": flag("KGE_RESCAL_FUSED"),
                "rescal_unfused": flag("KGE_RESCAL_UNFUSED"), "pull_dir": flag("KGE_PULL_DIR"),
                "transx_own": flag("KGE_TRANSX_OWN"), "own_staged": flag("KGE_OWN_STAGED"), "rescal_staged": flag("KGE_RESCAL_STAGED"),
                "dp_sparse": flag("KGE_DP_SPARSE"), "opt_rider": flag("KGE_OPT_RIDER"),
                "dp_allreduce": flag("KGE_DP_ALLREDUCE")}

    def _init_hot_path(self, process_group=None, backend=None, use_graph=None):
        # `backend` exists so that the multi-process plumbing (batch sharding, gradient collectives, replica
        # consistency) can be exercised on CPU/gloo with a checker injected by tests; the product default -- and the
        # only backend this package contains -- is the HIP library, which raises without a GPU.
        self.K = backend if backend is not None else K
        self.flat = None
        self.process_group = process_group
        self.use_graph = use_graph
        self.switches = self._switches()
        self._touched, self._touch_parity, self._touched_step = None, 0, None   # RESCAL: bitmaps of entity rows with a gradient
        self._gather_work = None      # N > 1: the in-flight all-gather of the previous step's parameters (see _reduce_and_step)
        self.phase_marks = None       # bench.py at N > 1: a list that receives (phase name, event) at the phase boundaries of a step
        self._graph = None
        self.world_size = 1
        self.rank = 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world_size = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        # an explicitly given process group is honoured even when it has ONE rank: the step then runs the same collectives
        # (degenerate, but every RCCL call and the multi-rank graph capture execute -- tests/test_hip_dist.py does this on
        # the one GPU a test box has)
        self.distributed = self.world_size > 1 or process_group is not None

    # ------------------------------------------------------------------ build
    def build_model(self, monitor=Monitor.FILTERED_MEAN_RANK):
        if self.config.optimizer not in K.OPTIMIZER_IDS:  # sgd / adam / adagrad / rms (utils/trainer.py:112-131)
            raise NotImplementedError("No support for %s optimizer" % self.config.optimizer)
        self.model.to(self.config.device)
        self._sparse_dp = self._sparse_dp_wanted()
        self._dp_allreduce = (not self._sparse_dp) and self._dp_allreduce_wanted()
        self.flat = FlatState(self.model, self.config.optimizer, self.K, self.world_size, self.rank, self.distributed,
                              replicate_optimizer=self._sparse_dp or self._dp_allreduce)
        self.evaluator = Evaluator(self.model, self.config, backend=self.K)
        self.loss_buf = self.K.new_loss_buffer(self.flat.param.device)
        self.monitor = monitor
        if self.early_stopper is None:   # utils/trainer.py:143 (config.patience is hard-wired to 3 there, config.py:55)
            self.early_stopper = PatienceStopper(getattr(self.config, "patience", -1), monitor)
        self.best_metric = None
        self._desc = self.K.model_desc(self.model, [v for v in self.flat.views], self.flat.grad_views)
        self._selfadv_ws = None
        if self.distributed:  # replicas must start identical
            torch.distributed.broadcast(self.flat.param, src=0, group=self.process_group)

    def _mark(self, name):
        """Per-phase timing of the multi-GPU step (bench.py --gpus N): an event on the current stream at a phase boundary.  The
        collectives are issued synchronously with respect to the stream (the stream waits for RCCL's), so an event behind a
        collective completes when the collective has."""
        if self.phase_marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.phase_marks.append((name, ev))

    def _wait_gather(self):
        """The previous step's parameter all-gather was issued asynchronously (N > 1, eager steps): everything that does not read
        the tables -- the next batch's sampler launch, host-side argument marshalling -- runs under it; the first reader waits here."""
        if self._gather_work is not None:
            self._gather_work.wait()
            self._gather_work = None

    # ------------------------------------------------------------------ one step (gradients into flat.grad)
    def _accumulate_pairwise(self, ph, pr, pt, nh, nr, nt, sampled=False):
        name = self.model.model_name.lower()
        if name == "rescal" and not getattr(self, "_rescal_normalised", False):
            # Rescal.embed renormalises both tables in place on every forward (pairwise.py:843-844).  Inside an epoch the
            # previous step's optimiser already stored them renormalised (_reduce_and_step): then this pass is skipped.
            self.K.rescal_normalize(self.flat.views[0], self.flat.views[1], self.model.hidden_size)
        if sampled and self.model.kernel_name == "transr" and nr.numel() == pr.numel():
            nr = pr      # (as for RESCAL below: pairs grouped by relation, the two-launch step of csrc/kge_transr_rows.hip)
        if sampled and name == "rescal" and nr.numel() == pr.numel():
            nr = pr      # our sampler corrupts heads and tails only: passing the SAME buffer lets kge_train_pairwise_hinge group
                         # pairs by relation and run the whole step in one launch (k_rescal_pair)
            if self._rescal_fused() and self.K.rescal_pair_step_ok(self._desc, ph.numel()) and not self.switches.get("rescal_unfused"):
                # ... which also marks the entity rows it writes a gradient into, so that the row-owner optimiser of this step
                # (_reduce_and_step) reads the gradient of those rows only.  Two bitmaps alternate with the step parity: a
                # step's optimiser resets the other one.  (A parity that repeats after an epoch boundary only leaves stale
                # bits behind: rows read needlessly, never a gradient missed.)
                par = self._touch_parity
                st = self._rescal_stage_for(ph.numel())
                if st is not None:   # entity gradient rows staged per (pair, side) and summed by the row owners: no float atomics
                    self.K.rescal_pair_step_staged(self._desc, ph, pr, pt, nh, nt, self.config.margin, self.loss_buf, self._touched_bitmaps()[par], st)
                else:
                    self.K.rescal_pair_step(self._desc, ph, pr, pt, nh, nt, self.config.margin, self.loss_buf, touched=self._touched_bitmaps()[par])
                self._touched_step, self._stage_step = par, st
                return
        if name == "rotate":
            self._selfadv_ws = self.K.train_pairwise_selfadv(self._desc, ph, pr, pt, nh, nr, nt, self.config.neg_rate,
                                                        self.config.alpha, self.loss_buf, self._selfadv_ws)
        else:
            self.K.train_pairwise_hinge(self._desc, ph, pr, pt, nh, nr, nt, self.config.margin, self.loss_buf)
            if self.model.kernel_name == "ntn":  # + get_reg(None, None, None) (utils/trainer.py:155): dense L2-norm
                # (one regulariser per GLOBAL step: the hinge gradients are summed over ranks)
                self.K.l2norm_reg(self.flat.param, self.flat.grad, self.model.lmbda / self.world_size, self.loss_buf)

    def _fused_sampler_ok(self):
        """Sampler + scoring + hinge + backward in one kernel: gather-type pairwise-hinge models with neg_rate 1."""
        return (self.K is K and self.model.training_strategy == TrainingStrategy.PAIRWISE_BASED
                and self.model.kernel_name not in ("rescal", "ntn", "transr") and self.model.model_name.lower() != "rotate"
                and int(self.config.neg_rate) == 1)

    def _fused_pointwise_ok(self):
        """Sampler + scoring + logistic loss + backward in one kernel for the pointwise models (a bundle's negatives
        must fit one lane group)."""
        if not (self.K is K and self.model.training_strategy == TrainingStrategy.POINTWISE_BASED):
            return False
        group = 32 if self.model.hidden_size <= 256 else 64
        return 1 <= int(self.config.neg_rate) <= group

    def _fused_rotate_ok(self, staged=False):
        """RotatE self-adversarial step with the sampler fused in (negatives of a positive must fit one lane group).  Rows of
        more than 1024 floats exist only in the staged form (four waves per bundle, csrc/kge_score.hip); the atomic form of
        such a model takes the stand-alone sampler + the explicit-id bundle kernel."""
        if not (self.K is K and self.model.model_name.lower() == "rotate" and self.model.kernel_name == "rotate"):
            return False
        if self.model.hidden_size > 1024 and not staged:
            return False
        group = 32 if self.model.hidden_size <= 256 else 64
        return int(self.config.neg_rate) <= group

    def _accumulate_next_batch(self, cursor=None, fixed_range=None):
        """One batch from the generator's stream into the gradient / loss buffers."""
        gen = self.generator
        self._step_cursor = cursor     # hipGraph-captured steps: the device-resident step state (element 2 = the step index)
        if self._fused_rotate_ok():
            start, n, offset = fixed_range if fixed_range is not None else gen._next_range()
            self._wait_gather()
            K.train_pairwise_selfadv_sampled(self._desc, gen.triples, gen.perm, start, n, gen.neg_rate, self.config.alpha,
                                             gen.bern, gen.slots, gen.seed, offset, self.loss_buf, cursor=cursor)
            return
        if self._fused_pointwise_ok():
            start, n, offset = fixed_range if fixed_range is not None else gen._next_range()
            self._wait_gather()
            K.train_pointwise_logistic_sampled(self._desc, gen.triples, gen.perm, start, n, gen.neg_rate, gen.bern, gen.slots,
                                               gen.seed, offset, self.model.kernel_lmbda(), self.model.kernel_reg_type(),
                                               self.loss_buf, cursor=cursor)
            return
        if self._fused_sampler_ok():
            start, n, offset = fixed_range if fixed_range is not None else gen._next_range()
            self._wait_gather()
            K.train_pairwise_hinge_sampled(self._desc, gen.triples, gen.perm, start, n, gen.bern, gen.slots, gen.seed,
                                           offset, self.config.margin, self.loss_buf, cursor=cursor)
            return
        if fixed_range is not None:
            pointwise = self.model.training_strategy != TrainingStrategy.PAIRWISE_BASED
            data = K.sample_batch(gen.triples, gen.perm, fixed_range[0], fixed_range[1], gen.neg_rate, self.config.tot_entity,
                                  gen.bern, gen.slots, gen.seed, fixed_range[2], pointwise=pointwise, out=self._sbuf,
                                  cursor=cursor)
        else:
            data = next(gen)
        self._wait_gather()      # (the sampler launch above ran under the previous step's parameter all-gather)
        if self.model.training_strategy == TrainingStrategy.PAIRWISE_BASED:
            if getattr(self, "_sparse_dp", False):       # the entity rows this rank's share of the batch can touch
                self._batch_entity_ids = torch.cat([data[0], data[2], data[3], data[5]])
            self._accumulate_pairwise(*data, sampled=True)
        else:
            if getattr(self, "_sparse_dp", False):
                self._batch_entity_ids = torch.cat([data[0], data[2]])
            self._accumulate_pointwise(*data)

    def _accumulate_pointwise(self, h, r, t, y):
        # rows arrive as bundles [positive, its neg_rate negatives] (generator / data/generator.py:125-156)
        self.K.train_pointwise_logistic(self._desc, h, r, t, y, self.model.kernel_lmbda(), self.model.kernel_reg_type(),
                                        self.loss_buf, bundle=1 + int(self.config.neg_rate))

    # ------------------------------------------------------------------ owner-computes ("pull") step: big TransE batches
    def _pull_ok(self):
        """TransE / TransM + pairwise hinge with neg_rate 1, single GPU, full batches too large for the launch-bound graph path:
        the whole step (sampling, scoring, hinge, backward, dense optimiser) runs without atomics or a gradient buffer
        and is bit-reproducible (csrc/kge_pull.hip).  KGE_PULL=0 / 1 overrides the batch-size rule."""
        if not (self.K is K and not self.distributed and self._pull_shape_ok()
                and self.generator is not None and self.generator.n_train >= self.config.batch_size):
            return False
        if self.switches["pull"] is not None:
            return self.switches["pull"]
        # one launch per step, enqueued by a native loop: faster than the hipGraph-replayed atomic step at every batch size
        # measured (FB15k shape: B=128 13.1 vs 16.5 us, B=4096 17.2 vs 24.6 us, B=32768 34.5 vs 60 us).  The limit is the
        # per-batch incidence index (batches that touch a small part of the tables list only those rows + a bitmap of all rows).
        return self._owner_index_fits(K.pull_groups_per_block(self.model.hidden_size), getattr(self.generator, "pull_segment", None))

    def _owner_index_fits(self, groups_per_block, segment=None, compact=None):
        """Does the incidence index of every batch of the epoch (built once on the device, kept resident) fit the budget?  The
        footprint is exact (kge_pull_index_geometry's strides: the builder sizes every batch for the worst case) and includes the
        two sampler list sets and the build's transient workspace.  A graph of a million entities stepped at the reference's
        default B = 128 would need tens of GB of per-batch row bitmaps: such runs take the index-free atomic step instead."""
        from .generator import PullIndex
        key = (groups_per_block, segment, compact, int(self.config.batch_size), self.generator.n_train)
        cache = self.__dict__.setdefault("_fit_cache", {})
        if key not in cache:
            B = int(self.config.batch_size)
            per = B // self.world_size if self.distributed and B % self.world_size == 0 else B
            n_batches = self.generator.n_train // B
            resident, peak = PullIndex.footprint(K, n_batches, per, self.config.tot_entity, self.config.tot_relation, segment,
                                                 groups_per_block, compact)
            budget = self.PULL_INDEX_BUDGET
            dev = self.flat.param.device if self.flat is not None else None
            if dev is not None and dev.type == "cuda":
                budget = min(budget, torch.cuda.mem_get_info(dev)[0] // 4)
            cache[key] = peak <= budget
            self._index_footprint = {"resident_bytes": resident, "peak_bytes": peak, "budget_bytes": budget, "fits": cache[key]}
        return cache[key]

    def _pull_shape_ok(self):
        """Model / shape conditions of the owner-computes kernels: TransE / TransM hinge with neg_rate 1, rows of float4s, and
        padded normalised tables the kernel's 32-bit gather offsets can address (csrc/kge_pull.hip: launch_pull_step)."""
        if not (self.model.kernel_name in ("transe", "transm") and self.model.hidden_size % 4 == 0 and self.model.hidden_size <= 1024
                and self.model.training_strategy == TrainingStrategy.PAIRWISE_BASED and int(self.config.neg_rate) == 1):
            return False
        rows = max(int(self.config.tot_entity), int(self.config.tot_relation))
        return rows * K.pull_partial_stride(self.model.hidden_size) * 4 <= 0xFFFFFFFF

    def _pull_dp_ok(self):
        """Data-parallel ranks: the local gradient by owner-computes (kge_pull_step in KGE_OPT_GRADIENT mode: every row of the
        flat gradient written once, no atomics, no clearing pass), then the sharded reduce-scatter / optimiser / all-gather
        step.  Same conditions as the single-GPU pull step, on the rank's share of the batch."""
        B, N = int(self.config.batch_size), self.world_size
        if not (self.K is K and self.distributed and self._pull_shape_ok()
                and self.generator is not None and B % N == 0 and self.generator.n_train >= B):
            return False
        if self.switches["pull"] is not None:
            return self.switches["pull"]
        return (B // N) * 2 > self.GRAPH_MAX_ROWS

    def _pull_dp_step(self):
        gen, cfg = self.generator, self.config
        idx = gen.pull_index()
        if getattr(self, "_pull", None) is None or not self._pull.grad_only:
            self._pull = PullState(self.flat, self.model, idx.batch_size, idx.max_slots, grad_only=True)
            self._pull.refresh_norms()
        ps = self._pull
        b = gen._batch_idx
        start, n, offset = gen._next_range()
        if b >= idx.n_batches or n != idx.batch_size:     # the short last batch: atomic-scatter kernels on this one
            ps.ready = None
            self.flat.grad.zero_()   # the gradient-mode steps leave the previous (reduced) gradient behind; the fallback ADDS
            self._accumulate_next_batch(fixed_range=(start, n, offset))
            self._reduce_and_step()
            ps.refresh_norms()
            return
        pairs, inc, items, multi = idx.batch(b)
        cur = ps.cur_list
        self._mark("begin")
        if ps.ready != (b, offset):
            ps.lists[cur].clear()
            K.pull_sample(pairs, idx.inv(b), cfg.tot_entity, gen.bern, gen.slots, gen.seed, offset, ps.lists[cur])
        nxt = None
        if gen._pending > 0 and b + 1 < idx.n_batches:   # the next batch's sampler rides in this launch
            per = idx.batch_size
            off_next = gen._draws + self.rank * per * gen.neg_rate
            nxt = (idx.batch(b + 1)[0], idx.inv(b + 1), gen.bern, gen.slots, gen.seed, off_next, ps.lists[cur ^ 1])
        key = (b, cur, nxt is not None)
        call = ps.calls.get(key)
        if call is None:   # arguments marshalled once per (batch, list set); only the ride-along Philox offset changes per epoch
            call = ps.calls[key] = K.pull_step(self._desc, ps.tables[1], ps.hats[0], None, ps.norms[0], None, None, None, pairs,
                                               ps.lists[cur], items, inc, ps.partials, multi, cfg.margin, "gradient", 0.0, 1,
                                               self.loss_buf, sample_next=nxt, dense_skip=idx.skip(b), prepare_only=True)
        call(nxt[5] if nxt is not None else None)
        if nxt is not None:
            ps.cur_list ^= 1
            ps.ready = (b + 1, nxt[5])
        else:
            ps.ready = None
        self._reduce_and_step(clear_local_grad=False)   # every row of the local gradient is rewritten by the next step
        ps.refresh_norms()
        self._mark("row_norms")

    def _pull_fixed_tables(self):
        """Non-trainable descriptor tables after the two embedding tables (TransM: the per-relation weights theta)."""
        return [self.model.theta.to(self.flat.param.device).contiguous()] if self.model.kernel_name == "transm" else []

    def _pull_two_phase(self):
        """The owner-computes step in two launches: every pair evaluated once, the owners sum the pairs' records
        (csrc/kge_pull.hip: k_pull_eval + k_pull_step<..., DIR>).  KGE_PULL_DIR=0 / 1 overrides."""
        if not bool(getattr(self.model, "l1_flag", False)):
            return False
        if self.switches.get("pull_dir") is not None:
            return self.switches["pull_dir"]
        # measured (profiles/r03_experiments.md section 11): L1, B = 32768: 35.0 -> 30.3 us per step (TransM 47.4 -> 44.9);
        # B = 16384: 31.5 -> 30.8, B = 8192: 27.3 -> 23.7, B = 4096: 19.1 -> 20.4, B = 128: 13.0 -> 15.6 (a second launch costs
        # more than the re-evaluations it saves);
        # L2 has no two-phase form: staged residual rows (46.3 -> 53.6 us) and the scalar-record form whose owners recompute the
        # residuals from their gathers (45.9 -> 51.3 us, profiles/r05_l2_mid_ab.txt) both lost to the one-phase step and were removed
        return bool(getattr(self.model, "l1_flag", False)) and int(self.config.batch_size) >= self.PULL_TWO_PHASE_MIN_BATCH

    def _pull_state(self):
        idx = self.generator.pull_index()
        if getattr(self, "_pull", None) is None or self._pull.batch_size != idx.batch_size:
            ps = self._pull = PullState(self.flat, self.model, idx.batch_size, idx.max_slots, two_phase=self._pull_two_phase())
            ps.sync_in()
            gen, cfg = self.generator, self.config
            ps.plan = K.PullPlan(self.model.kernel_name, self.model.desc_kwargs(), cfg.tot_entity, cfg.tot_relation,
                                 [t + self._pull_fixed_tables() for t in ps.tables], ps.hats,
                                 ps.norms, ps.state1, ps.state2, ps.lists, idx, ps.partials, cfg.margin, cfg.optimizer,
                                 cfg.learning_rate, self.loss_buf, gen.bern, gen.slots, gen.seed,
                                 idx.batch_size * gen.neg_rate, direction=ps.direction)
        return self._pull, idx

    def _pull_steps(self, n_steps):
        """The next n_steps steps of the current epoch, enqueued by one native call (kge_pull_run)."""
        ps, idx = self._pull_state()
        gen = self.generator
        B = idx.batch_size
        if n_steps <= 0:
            return
        first = gen._batch_idx
        if gen._pending < n_steps or first + n_steps > idx.n_batches:
            raise StopIteration
        offset = gen._draws
        gen._batch_idx += n_steps
        gen._pending -= n_steps
        gen._draws += n_steps * B * gen.neg_rate
        ready = ps.ready == (first, offset)
        if not ready and ps.ready is not None:   # a sampler rode along for a batch that is not the next one: discard
            ps.lists[ps.cur_list].clear()
        # the last step carries the sampler of the following batch: the next one of this epoch (1), or -- at the end of the
        # epoch -- batch 0 of the next epoch (2: the permutation is walked again and the Philox counters keep counting, so the
        # draw is exactly the one the next epoch's first step would make; if no epoch follows, the unused set is discarded by
        # the `ready` check above).  Saves the stand-alone sampler launch (12 us at B = 32768) at every epoch boundary.
        after = 1 if (gen._pending > 0 and first + n_steps < idx.n_batches) else (2 if gen._pending == 0 else 0)
        ps.plan.run(first, n_steps, ps.cur, ps.cur_list, ready, self.flat.step + 1, offset, after)
        self.flat.step += n_steps
        carried = n_steps - 1 + (1 if after else 0)            # number of ride-along samplers = list-set flips
        ps.cur ^= n_steps & 1
        ps.cur_list ^= carried & 1
        next_batch = first + n_steps if after == 1 else 0
        ps.ready = (next_batch, offset + n_steps * B * gen.neg_rate) if after else None

    def pull_step_explicit(self, ph, pr, pt, nh, nr, nt, segment=None, compact=None):
        """The owner-computes step on an explicit batch (positives + given negatives, neg_rate 1): the incidence index
        of this one batch is built on the host first, so this is for parity tests and one-off batches, not the hot loop."""
        import numpy as np
        from .generator import PullIndex
        pos = np.stack([x.detach().cpu().numpy() for x in (ph, pr, pt)], 1)
        idx = PullIndex([pos], self.config.tot_entity, self.config.tot_relation, self.flat.param.device, segment,
                        K.pull_groups_per_block(self.model.hidden_size), compact=compact)
        ps = PullState(self.flat, self.model, len(pos), idx.max_slots, two_phase=self._pull_two_phase())
        ps.sync_in()
        pairs, inc, items, multi = idx.batch(0)
        skip = idx.skip(0)
        K.pull_lists_explicit(pairs, idx.inv(0), nh.contiguous(), nt.contiguous(), ps.lists[0])
        self.flat.step += 1
        desc = K.make_desc(self.model.kernel_name, ps.tables[0] + self._pull_fixed_tables(), None, tot_entity=self.config.tot_entity,
                           tot_relation=self.config.tot_relation, **self.model.desc_kwargs())
        K.pull_step(desc, ps.tables[1], ps.hats[0], ps.hats[1], ps.norms[0], ps.norms[1], ps.state1, ps.state2, pairs,
                    ps.lists[0], items, inc,
                    ps.partials, multi, self.config.margin, self.config.optimizer, self.config.learning_rate,
                    self.flat.step, self.loss_buf, dense_skip=skip, direction=ps.direction)
        ps.cur = 1
        ps.sync_out()

    # ------------------------------------------------------------------ two-phase owner-computes step: DistMult / ComplEx
    def _own_ok(self):
        """DistMult / ComplEx(N3) + pointwise logistic with neg_rate 1 on one GPU: the whole step without float atomics in two
        launches (csrc/kge_own.hip: gradient rows by their owners, then the in-place optimiser on the touched rows), the epoch
        enqueued by one native call.  KGE_PW_PULL=0 / 1 overrides."""
        m = self.model
        if not (self.K is K and not self.distributed and m.training_strategy == TrainingStrategy.POINTWISE_BASED
                and int(self.config.neg_rate) == 1
                and self.generator is not None and self.generator.n_train >= self.config.batch_size):
            return False
        if m.kernel_name in ("distmult", "complex"):          # the specialised kernels (csrc/kge_own.hip)
            if not (m.hidden_size % 4 == 0 and m.hidden_size <= 512 and len({p.weight.shape[1] for p in m.parameter_list}) == 1
                    and m.kernel_reg_type() in (0, 1, 2, 3)):
                return False
        elif m.kernel_name in self.OWN_GENERIC_MODELS:        # the model-generic staged step (csrc/kge_ownx.hip)
            if not (K.own_groups_per_block(m.kernel_name, m.hidden_size) > 0 and self.switches.get("own_staged") is not False):
                return False
        else:
            return False
        if self.switches["pw_pull"] is not None:
            return self.switches["pw_pull"]
        if self.switches["staged"] is not None:    # an explicit KGE_STAGED=0 / 1 asks for the atomic / staged A/B pair
            return False
        # Measured against the (hipGraph-replayed) atomic-scatter step with the staged form (profiles/r03_experiments.md section 14):
        # it wins at every batch size and optimiser tried -- DistMult FB15k B = 128 / 1024 / 4096 / 8192 / 32768: 16.8 / 19.7 / 32.4
        # / 50.2 / 70 -> 14.3 / 17.6 / 23.6 / 29.8 / 57.6 us; ComplEx WN18RR B = 128 Adagrad 39.9 -> 16.1, Adam 94.5 -> 70.3 us.
        # -- as long as the per-batch index fits (the same guard as the TransE path)
        return self._owner_index_fits(K.own_groups_per_block(m.kernel_name, m.hidden_size), None, None if self._own_dense() else True)

    def _own_dense(self):
        return self.config.optimizer in ("adam", "rms")   # optimisers that move every row every step

    def _own_state(self):
        gen, cfg, flat = self.generator, self.config, self.flat
        name = self.model.kernel_name
        # sparse optimisers only ever visit touched rows: always the compact (touched rows + bitmap) index
        idx = gen.pull_index(groups_per_block=K.own_groups_per_block(name, self.model.hidden_size),
                             compact=None if self._own_dense() else True)
        st = getattr(self, "_own", None)
        if st is None or st["index"] is not idx:
            dev = flat.param.device
            generic = name in self.OWN_GENERIC_MODELS     # (csrc/kge_ownx.hip: owners apply in place, no gradient rows are kept)
            gbuf = None if generic else torch.empty_like(flat.param)      # gradient rows (only the touched ones are ever written / read)
            off = [v.data_ptr() - flat.param.data_ptr() for v in flat.views]
            view = lambda buf: [buf[o // 4:o // 4 + v.numel()].view_as(v) for o, v in zip(off, flat.views)]
            desc = self.model.make_desc(flat.views, None if generic else view(gbuf))
            s1 = view(flat.state1) if flat.state1 is not None else None
            s2 = view(flat.state2) if flat.state2 is not None else None
            lists = [K.PullListSet(idx.batch_size, cfg.tot_entity, dev) for _ in range(2)]
            stride = K.own_partial_stride(name, self.model.hidden_size)
            partials = torch.empty(max(1, idx.max_slots) * stride, dtype=torch.float32, device=dev)
            # staged form (default; KGE_OWN_STAGED=0: owners re-evaluate): every bundle evaluated once, its gradient rows left in
            # four slots per bundle for their owners to add
            staged = self.switches.get("own_staged")
            stage = (torch.empty(K.own_stage_floats(name, self.model.hidden_size, idx.batch_size), dtype=torch.float32, device=dev)
                     if (staged is None or staged) else None)
            plan = K.OwnPlan(desc, s1, s2, lists, idx, partials, cfg.optimizer, cfg.learning_rate, self.model.kernel_lmbda(),
                             self.model.kernel_reg_type(), self.loss_buf, gen.bern, gen.slots, gen.seed, idx.batch_size * gen.neg_rate,
                             stage=stage)
            st = self._own = dict(index=idx, gbuf=gbuf, desc=desc, s1=s1, s2=s2, lists=lists, partials=partials, plan=plan,
                                  cur_list=0, ready=None, stage=stage)
        return st

    def _own_steps(self, n_steps):
        """The next n_steps steps of the current epoch, enqueued by one native call (kge_own_run)."""
        if n_steps <= 0:
            return
        st = self._own_state()
        gen, idx = self.generator, st["index"]
        B = idx.batch_size
        first = gen._batch_idx
        if gen._pending < n_steps or first + n_steps > idx.n_batches:
            raise StopIteration
        offset = gen._draws
        gen._batch_idx += n_steps
        gen._pending -= n_steps
        gen._draws += n_steps * B * gen.neg_rate
        ready = st["ready"] == (first, offset)
        if not ready and st["ready"] is not None:   # a sampler rode along for a batch that is not the next one: discard
            st["lists"][st["cur_list"]].clear()
        # (as _pull_steps: 1 = the next batch of this epoch, 2 = batch 0 of the next epoch rides in the epoch's last step)
        after = 1 if (gen._pending > 0 and first + n_steps < idx.n_batches) else (2 if gen._pending == 0 else 0)
        st["plan"].run(first, n_steps, st["cur_list"], ready, self.flat.step + 1, offset, after)
        self.flat.step += n_steps
        carried = n_steps - 1 + (1 if after else 0)
        st["cur_list"] ^= carried & 1
        st["ready"] = ((first + n_steps if after == 1 else 0), offset + n_steps * B * gen.neg_rate) if after else None

    def own_step_explicit(self, h, r, t, y):
        """The two-phase step on an explicit pointwise batch in the sampler's layout for neg_rate 1 (rows 2i = positive i, 2i+1 = its
        corruption): the incidence index of this one batch is built on the host first -- parity tests and one-off batches."""
        import numpy as np
        from .generator import PullIndex
        name, cfg, flat = self.model.kernel_name, self.config, self.flat
        hh, rr, tt, yy = (x.detach().cpu().numpy() for x in (h, r, t, y))
        if len(hh) % 2 or not (np.all(yy[0::2] == 1) and np.all(yy[1::2] == -1) and np.array_equal(rr[0::2], rr[1::2])):
            raise ValueError("own_step_explicit: rows must alternate positive / its corruption (neg_rate 1 pointwise layout)")
        pos = np.stack([hh[0::2], rr[0::2], tt[0::2]], 1)
        dense = self._own_dense()
        idx = PullIndex([pos], cfg.tot_entity, cfg.tot_relation, flat.param.device, None,
                        K.own_groups_per_block(name, self.model.hidden_size), compact=None if dense else True)
        pairs, inc, items, multi = idx.batch(0)
        dev = flat.param.device
        lists = K.PullListSet(len(pos), cfg.tot_entity, dev)
        K.pull_lists_explicit(pairs, idx.inv(0), h[1::2].contiguous(), t[1::2].contiguous(), lists)
        gbuf = torch.empty_like(flat.param)
        off = [v.data_ptr() - flat.param.data_ptr() for v in flat.views]
        view = lambda buf: [buf[o // 4:o // 4 + v.numel()].view_as(v) for o, v in zip(off, flat.views)]
        gviews = view(gbuf)
        desc = K.make_desc(name, flat.views, gviews, tot_entity=cfg.tot_entity, tot_relation=cfg.tot_relation, **self.model.desc_kwargs())
        stride = K.own_partial_stride(name, self.model.hidden_size)
        partials = torch.empty(max(1, idx.max_slots) * stride, dtype=torch.float32, device=dev)
        staged = self.switches.get("own_staged")
        stage = (torch.empty(K.own_stage_floats(name, self.model.hidden_size, len(pos)), dtype=torch.float32, device=dev)
                 if (staged is None or staged) else None)
        if name in self.OWN_GENERIC_MODELS:     # the model-generic step exists as a run only: a one-batch plan, lists given
            desc = self.model.make_desc(flat.views, gviews)
            plan = K.OwnPlan(desc, view(flat.state1) if flat.state1 is not None else None,
                             view(flat.state2) if flat.state2 is not None else None, [lists, K.PullListSet(len(pos), cfg.tot_entity, dev)],
                             idx, partials, cfg.optimizer, cfg.learning_rate, self.model.kernel_lmbda(), self.model.kernel_reg_type(),
                             self.loss_buf, None, None, 0, len(pos), stage=stage)
            flat.step += 1
            plan.run(0, 1, 0, True, flat.step, 0, 0)
            return None
        K.own_step(desc, pairs, lists, items, idx.skip(0), inc, partials, dense, self.model.kernel_lmbda(), self.model.kernel_reg_type(),
                   self.loss_buf, reset_lists=False, stage=stage)
        flat.step += 1
        K.own_apply(desc, view(flat.state1) if flat.state1 is not None else None, view(flat.state2) if flat.state2 is not None else None,
                    pairs, lists, items, idx.skip(0), multi, partials, dense, cfg.optimizer, cfg.learning_rate, flat.step)
        return gviews

    # ------------------------------------------------------------------ TransH / TransD: gradients without float atomics
    TRANSX_OWN_MIN_BATCH = 1

    def _transx_ok(self):
        """TransH / TransD hinge step with neg_rate 1 on one GPU at large batches: every pair evaluated once with its gradient
        rows staged, one owner per parameter row sums them into the dense gradient tables (csrc/kge_pullx.hip), then the flat
        optimiser.  No float atomics, bit-reproducible.  KGE_TRANSX_OWN=0 / 1 overrides the batch-size rule."""
        m = self.model
        if not (self.K is K and not self.distributed and m.kernel_name in ("transh", "transd")
                and m.training_strategy == TrainingStrategy.PAIRWISE_BASED and int(self.config.neg_rate) == 1
                and len({p.weight.shape[1] for p in m.parameter_list}) == 1
                and m.parameter_list[0].weight.shape[1] % 4 == 0 and m.parameter_list[0].weight.shape[1] <= 512
                and self.generator is not None and self.generator.n_train >= self.config.batch_size):
            return False
        if self.switches.get("transx_own") is not None:
            return self.switches["transx_own"]
        # measured against the hipGraph-replayed atomic step at FB15k shape (profiles/r03_experiments.md section 14): TransH B = 128 /
        # 1024 / 4096 / 8192 / 32768: 27.3 / 29.0 / 31.2 / 36.8 / 80 -> 19.8 / 22.5 / 27.9 / 35.4 / 64 us; TransD 1024 / 4096 / 8192 /
        # 32768: 40.0 / 42.6 / 51.5 / 117 -> 30.1 / 35.7 / 43.4 / 84 us
        if int(self.config.batch_size) < self.TRANSX_OWN_MIN_BATCH:
            return False
        return self._owner_index_fits(K.transx_groups_per_block(m.parameter_list[0].weight.shape[1]), 32)

    def _transx_state(self):
        st = getattr(self, "_transx", None)
        d = self.model.parameter_list[0].weight.shape[1]
        # (compact = None: the index lists only the touched rows when the batch touches a small part of the tables, else every row)
        idx = self.generator.pull_index(groups_per_block=K.transx_groups_per_block(d), segment=32)
        if st is None or st["index"] is not idx:
            dev = self.flat.param.device
            E = int(self.config.tot_entity)
            st = self._transx = {
                "index": idx, "lists": [K.PullListSet(idx.batch_size, E, dev) for _ in range(2)], "cur_list": 0, "ready": None,
                "partials": torch.empty(max(1, idx.max_slots) * K.transx_partial_stride(d), dtype=torch.float32, device=dev),
                "scratch": K.TransXScratch(self.model.kernel_name, d, idx.batch_size, dev), "plan": None}
        return st

    def _transx_steps(self, n_steps):
        """The next n_steps steps of the current epoch, enqueued by one native call (kge_transx_run: per step evaluate pairs /
        owners sum / finish, with the next batch's sampler riding along, then the dense optimiser over the flat buffers)."""
        st = self._transx_state()
        gen, cfg, idx = self.generator, self.config, st["index"]
        B = idx.batch_size
        if st.get("plan") is None:
            st["plan"] = K.TransXPlan(self._desc, self.flat, st["lists"], idx, st["partials"], st["scratch"], cfg.margin, cfg.optimizer,
                                      cfg.learning_rate, self.loss_buf, gen.bern, gen.slots, gen.seed, B * gen.neg_rate)
        first = gen._batch_idx
        if gen._pending < n_steps or first + n_steps > idx.n_batches:
            raise StopIteration
        offset = gen._draws
        gen._batch_idx += n_steps
        gen._pending -= n_steps
        gen._draws += n_steps * B * gen.neg_rate
        ready = st["ready"] == (first, offset)
        if not ready and st["ready"] is not None:   # a sampler rode along for a batch that is not the next one: discard
            st["lists"][st["cur_list"]].clear()
        after = 1 if (gen._pending > 0 and first + n_steps < idx.n_batches) else (2 if gen._pending == 0 else 0)
        st["plan"].run(first, n_steps, st["cur_list"], ready, self.flat.step + 1, offset, after)
        self.flat.step += n_steps
        carried = n_steps - 1 + (1 if after else 0)
        st["cur_list"] ^= carried & 1
        st["ready"] = ((first + n_steps if after == 1 else 0), offset + n_steps * B * gen.neg_rate) if after else None

    def transx_step_explicit(self, ph, pr, pt, nh, nr, nt):
        """The same step on an explicit batch (positives + given negatives, neg_rate 1): gradients into the flat buffer, no
        optimiser (parity tests read them; the caller steps).  The incidence index of this one batch is built on the host."""
        import numpy as np
        from .generator import PullIndex
        d = self.model.parameter_list[0].weight.shape[1]
        dev = self.flat.param.device
        pos = np.stack([x.detach().cpu().numpy() for x in (ph, pr, pt)], 1)
        idx = PullIndex([pos], self.config.tot_entity, self.config.tot_relation, dev, 32, K.transx_groups_per_block(d), compact=True)
        lists = K.PullListSet(len(pos), int(self.config.tot_entity), dev)
        pairs, inc, items, multi = idx.batch(0)
        K.pull_lists_explicit(pairs, idx.inv(0), nh.contiguous(), nt.contiguous(), lists)
        partials = torch.empty(max(1, idx.max_slots) * K.transx_partial_stride(d), dtype=torch.float32, device=dev)
        scratch = K.TransXScratch(self.model.kernel_name, d, len(pos), dev)
        K.transx_grad_step(self._desc, pairs, lists, items, idx.skip(0), inc, partials, multi, self.config.margin, scratch, self.loss_buf)

    # ------------------------------------------------------------------ staged (atomic-free) step of the long-row bundle kernels
    def _staged_ok(self):
        """RotatE self-adversarial / DistMult / ComplEx logistic step without a gradient buffer: the bundle kernel stages its
        gradient rows with plain stores and the optimiser sweep sums them per parameter row in a fixed order (csrc/kge_staged.hip) -- no fp32
        atomics, bit-reproducible.  Single GPU, batches beyond the launch-bound graph regime; KGE_STAGED=0 / 1 overrides."""
        from .generator import StagedIndex
        if not (self.K is K and not self.distributed and self.generator is not None):
            return False
        if not (self._fused_rotate_ok(staged=True) or (self._fused_pointwise_ok() and self.model.kernel_name in ("distmult", "complex"))):
            return False
        dims = {p.weight.shape[1] for p in self.model.parameter_list}
        if len(dims) != 1 or self.model.hidden_size % 4 or self.model.hidden_size > 2048:
            return False
        nb = (self.generator.n_train + self.generator.batch_size - 1) // self.generator.batch_size
        if not StagedIndex.fits(nb, self.config.tot_entity, self.config.tot_relation):
            return False
        if self.switches["staged"] is not None:
            return self.switches["staged"]
        if self._own_ok():
            return False
        # default: the long-row RotatE bundles beyond the graph regime (C3: 320 -> 230 us per step).  For the pointwise models
        # the staged step is correct and deterministic but not faster than atomics + hipGraph replay at the measured shapes
        # (profiles/r02_experiments.md), so it stays opt-in (KGE_STAGED=1).
        return self.model.kernel_name == "rotate" and self.config.batch_size * (1 + int(self.config.neg_rate)) > self.GRAPH_MAX_ROWS

    def _staged_plan(self):
        if getattr(self, "_staged", None) is None:
            cfg, flat = self.config, self.flat
            rows = [p.weight.shape[0] for p in self.model.parameter_list]
            self._staged = K.StagedPlan(self.model.kernel_name, flat.param, flat.state1, flat.state2, flat.offsets, rows,
                                        self.model.hidden_size, cfg.tot_entity, cfg.tot_relation, cfg.batch_size,
                                        int(cfg.neg_rate), sparse=cfg.optimizer in ("sgd", "adagrad"))
        return self._staged

    def _staged_step(self):
        gen, cfg = self.generator, self.config
        b = gen._batch_idx
        start, n, offset = gen._next_range()
        if n <= 0:
            return
        plan = self._staged_plan()
        n_idx = plan.bind_batch(b, gen.staged_index())
        assert n_idx == n
        if self.model.kernel_name == "rotate":
            K.train_pairwise_selfadv_sampled_staged(self._desc, gen.triples, gen.perm, start, n, gen.neg_rate, cfg.alpha,
                                                    gen.bern, gen.slots, gen.seed, offset, plan, self.loss_buf)
        else:
            K.train_pointwise_logistic_sampled_staged(self._desc, gen.triples, gen.perm, start, n, gen.neg_rate, gen.bern,
                                                      gen.slots, gen.seed, offset, self.model.kernel_lmbda(),
                                                      self.model.kernel_reg_type(), plan, self.loss_buf)
        self.flat.step += 1
        K.optimizer_step_staged(cfg.optimizer, plan, cfg.learning_rate, self.flat.step)

    def sync_model(self):
        """After stepping: make the model's parameters (FlatState.param) hold the current tables."""
        if getattr(self, "_pull", None) is not None:
            self._pull.sync_out()

    def step_next_batch(self):
        """One whole training step on the generator's next batch, whichever path serves it."""
        self.step_next_batches(1)

    # The step paths, in order of precedence: the first whose predicate holds serves the next batches.  (name, predicate, needs
    # FULL batches -- the owner-computes runs index only the whole batches of the permutation; a short last batch falls through).
    #
    #   path      models                                   when (defaults; DESIGN.md section 5a lists the overriding switches)           step
    #   pull      TransE, TransM                           hinge, neg_rate 1, one GPU, d % 4 == 0 and <= 1024, index within budget       kge_pull_run: [k_pull_eval +] k_pull_step per step (two launches for L1 at B >= 8192)
    #   own       DistMult, ComplEx(N3), ANALOGY, CP,      pointwise logistic, neg_rate 1, one GPU, index within budget                  kge_own_run: k_own_eval + k_own_step (+ k_own_apply)
    #             SimplE(_ignr), QuatE
    #   staged    RotatE (DistMult / ComplEx: KGE_STAGED)  one GPU, a step scores more than GRAPH_MAX_ROWS triples                       bundle kernel with staged gradient rows + k_opt_staged
    #   transx    TransH, TransD                           hinge, neg_rate 1, one GPU, d % 4 == 0 and <= 512, index within budget        kge_transx_run: k_transx_eval + k_transx_own + k_opt
    #   pull_dp   TransE, TransM at N > 1                  batch per rank beyond the graph regime                                        k_pull_step<gradient> -> reduce-scatter -> sharded k_opt -> all-gather -> k_row_norms
    #   generic   everything else (RESCAL, NTN, TransR,    --                                                                            fused step kernel (atomic scatter) + k_opt / k_opt_rows4; replayed as a hipGraph
    #             neg_rate > 1, short last batches, N > 1)                                                                               when a step scores <= GRAPH_MAX_ROWS triples (train_model_epoch); at N > 1 the dense or
    #                                                                                                                                    the sparse-row gradient exchange (_reduce_and_step)
    STEP_PATHS = (("pull", "_pull_ok", True), ("own", "_own_ok", True), ("staged", "_staged_ok", False), ("transx", "_transx_ok", True),
                  ("pull_dp", "_pull_dp_ok", False), ("generic", None, False))

    def step_path(self, n=1):
        """Name of the path that serves the next n batches (STEP_PATHS)."""
        full = n > 0 and self.generator is not None and \
            self.generator._batch_idx + n <= self.generator.n_train // int(self.config.batch_size)
        for name, pred, needs_full in self.STEP_PATHS:
            if pred is None or ((full or not needs_full) and getattr(self, pred)()):
                return name

    def step_next_batches(self, n):
        """n consecutive steps of the current epoch.  On the owner-computes paths they are enqueued by one native call."""
        path = self.step_path(n)
        if path not in ("pull", "own") and getattr(self, "_pull", None) is not None and not self._pull.grad_only:
            # leaving the single-GPU pull path (a short last batch): hand the tables back.  (The gradient-mode state of the
            # data-parallel step owns no tables and is kept: its prepared calls and ride-along sampler survive.)
            self.sync_model()
            self._pull = None
        if path == "pull":
            self._pull_steps(n)
        elif path == "own":
            self._own_steps(n)
        elif path == "staged":
            for _ in range(n):
                self._staged_step()
        elif path == "transx":
            self._transx_steps(n)
        elif path == "pull_dp":
            for _ in range(n):
                self._pull_dp_step()
        else:
            try:
                for _ in range(n):
                    self._rescal_last = self.generator._pending == 1   # the epoch's last step leaves RESCAL's tables as the optimiser wrote them
                    self._mark("begin")
                    self._accumulate_next_batch()
                    self._reduce_and_step(overlap_gather=self.distributed)
            finally:
                self._wait_gather()

    def _touched_bitmaps(self):
        if getattr(self, "_touched", None) is None:
            words = (self.flat.views[0].shape[0] + 31) // 32
            buf = torch.zeros(2 * words, dtype=torch.int32, device=self.flat.param.device)
            self._touched = (buf[:words], buf[words:])
        return self._touched

    def _rescal_fused(self):
        """RESCAL's renormalisation folded into the optimiser launch: inside train_model_epoch, one GPU, the HIP backend, rows that
        fit the row-owner kernel.  KGE_RESCAL_FUSED=0 switches it off (A/B)."""
        if not (getattr(self, "_in_epoch", False) and self.K is K and not self.distributed and self.model.kernel_name == "rescal"
                and self.model.hidden_size <= 1024):
            return False
        if self.switches.get("rescal_fused") is not None:
            return self.switches["rescal_fused"]
        # pays once the entity table is a real stream (C4, 98.5 MB: 262 -> 251 us per step); on small tables the second optimiser
        # launch costs more than the pass it saves (FB15k preset, 3 MB: 64.5 -> 68.6 us)
        return self.flat.views[0].numel() * 4 >= (32 << 20)

    def _rescal_stage_for(self, n_pairs):
        """The staging buffers of the atomic-free entity gradients (kernels.RescalStage), or None when the step of n_pairs pairs cannot
        stage (large batches take the split kernels; hidden sizes that are not multiples of 4; KGE_RESCAL_STAGED=0; a test backend)."""
        if self.K is not K or self.switches.get("rescal_staged") is False or not hasattr(self.K, "rescal_stage_ok"):
            return None
        if not self.K.rescal_stage_ok(self._desc, n_pairs):
            return None
        st = getattr(self, "_rescal_stage", None)
        if st is None or st.n_pairs < n_pairs:
            ent = self.flat.views[0]
            st = self._rescal_stage = K.RescalStage(ent.shape[0], max(n_pairs, int(self.config.batch_size)), ent.shape[1], ent.device)
            self._graph = None     # (captured steps hold the old buffers' addresses)
            self._report_rescal_reproducibility()
        return st

    RESCAL_CHUNK_PAIRS = 64      # pairs of one relation a slab workgroup takes (csrc/kge_relgroup.h: kSlabChunk)

    def _report_rescal_reproducibility(self):
        """The staged RESCAL step is bit-reproducible while no relation has more than 64 pairs in a batch: a longer relation spans
        several chunks, whose shares of the relation-matrix gradient meet in float atomics (and its pairs keep their arrival order in
        the grouping).  Batches are fixed slices of one permutation, so which side of that line a run is on is known up front: computed
        once (one bincount over the epoch order), exposed as `rescal_reproducible`, and said out loud when False (hub
        relations of YAGO3-10 / FB15k put most real batches beyond 64 pairs)."""
        gen = self.generator
        self.rescal_reproducible = None
        if gen is None or getattr(gen, "perm", None) is None or getattr(gen, "triples", None) is None:
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():   # (the answer needs a device -> host read)
            return
        B, n, R = int(gen.batch_size), int(gen.n_train), int(self.config.tot_relation)
        nb = (n + B - 1) // B
        if nb * R > (1 << 27):       # (a 512 MB histogram is not worth the answer)
            return
        rel = gen.triples[gen.perm[:n], 1]
        key = (torch.arange(n, device=rel.device) // B) * R + rel
        worst = int(torch.bincount(key, minlength=nb * R).max().item())
        self.rescal_reproducible = worst <= self.RESCAL_CHUNK_PAIRS
        if not self.rescal_reproducible:
            _log("RESCAL staged step: a relation occurs %d times in one batch of %d (> %d): its share of the relation-matrix gradient is "
                 "summed with float atomics -- results are equal to rounding, not bit-reproducible, on this graph" % (worst, B, self.RESCAL_CHUNK_PAIRS))

    def _mean_type_loss(self):
        """pointwise_logistic and the self-adversarial loss are MEANS over the batch (criterion.py:13-23,31-34);
        the hinge is a SUM (criterion.py:25-29)."""
        return (self.model.training_strategy != TrainingStrategy.PAIRWISE_BASED
                or self.model.model_name.lower() == "rotate")

    # ------------------------------------------------------------------ data parallel: one all-reduce for small tables
    DP_ALLREDUCE_MAX_BYTES = 32 << 20

    def _dp_allreduce_wanted(self):
        """Data parallel with tables small enough that the optimiser sweep is a few microseconds (C0 / C1: 6.5 MB): ONE all-reduce of the
        flat gradient and the full dense optimiser on every rank, instead of reduce-scatter -> sharded optimiser -> all-gather.  The
        same bytes cross the links, but a step pays one collective's launch + synchronisation latency instead of two -- and at
        these sizes that latency, not the bytes, is the exchange.  Larger tables keep the sharded step, whose
        optimiser sweep and state shrink N-fold.  KGE_DP_ALLREDUCE=0/1 overrides."""
        if not self.distributed:
            return False
        if self.switches.get("dp_allreduce") is not None:
            return self.switches["dp_allreduce"]
        return sum(p.weight.numel() for p in self.model.parameter_list) * 4 <= self.DP_ALLREDUCE_MAX_BYTES

    # ------------------------------------------------------------------ data parallel: exchange only the rows the global batch touched
    # models whose step takes materialised batch ids (kge_sample_batch) and whose gradient is confined to the batch's rows (NTN's
    # dense L2 regulariser touches every row of every table: it keeps the dense exchange)
    SPARSE_DP_MODELS = ("rescal", "transr")

    def _sparse_dp_wanted(self):
        """Data parallel with a batch that touches a small part of the entity tables (C4: 4 x 1 024 of 123 182 rows): instead of
        reduce-scattering and all-gathering the dense flat buffers (2 x 104 MB on the wire at C4), the ranks exchange the gradient
        ROWS of the entities the global batch names -- all-gather of the id lists, one all-reduce of a [4 x global batch, d]
        buffer per entity table -- plus one dense all-reduce of the (small) remaining tables, and every rank then runs the same
        dense optimiser step over all rows (replicated; no parameter all-gather).  Replicas stay bit-identical (every rank sums
        the same all-reduced rows); at world size 2 the result equals the dense exchange bit for bit (tests/test_dist_gloo.py).
        Rule: the id lists' capacity (4 ids per pair of the GLOBAL batch) is at most a quarter of the entity rows.  KGE_DP_SPARSE=0/1."""
        if not self.distributed or self.model.kernel_name not in self.SPARSE_DP_MODELS:
            return False
        if self.switches.get("dp_sparse") is not None:
            return self.switches["dp_sparse"]
        rows = 4 * int(self.config.batch_size) * (1 + int(self.config.neg_rate)) // 2
        return 4 * rows <= int(self.config.tot_entity)

    def _sparse_exchange(self, mean):
        """Sum, over ranks, the gradient rows of the entity-indexed tables named by this step's batch ids, and the dense gradients of
        every other table.  All index work is device-side torch indexing on small tensors (no host synchronisation)."""
        dist = torch.distributed
        flat, E = self.flat, int(self.config.tot_entity)
        ids = self._batch_entity_ids.to(torch.int64)
        n_local = ids.numel()
        cap = n_local * self.world_size
        st = self.__dict__.setdefault("_sparse_bufs", {})
        if st.get("cap") != cap:
            dev = flat.param.device
            st.update(cap=cap, ids_all=torch.empty(cap, dtype=torch.int64, device=dev), mask=torch.zeros(E, dtype=torch.int32, device=dev),
                      bufs={})
        dist.all_gather_into_tensor(st["ids_all"], ids.contiguous(), group=self.process_group)
        ids_all = st["ids_all"]
        mask = st["mask"]
        mask.zero_()
        mask[ids_all] = 1
        slot_of_row = torch.cumsum(mask, 0) - 1          # the same on every rank: position of a touched row in the exchange buffer
        s_local, s_all = slot_of_row[ids], slot_of_row[ids_all]
        # which tables are looked up by ENTITY id is a property of the model (kernels._TABLE_SHAPES), not of a row count: with
        # tot_relation == tot_entity a relation table has E rows too, and reducing only its rows named by entity ids would let the
        # replicas drift apart silently
        roles = K._TABLE_SHAPES.get(self.model.model_name)
        by_entity = [i < len(roles) and roles[i][0] == "E" for i in range(len(flat.views))] if roles else [False] * len(flat.views)
        for (v, g), sparse in zip(zip(flat.views, flat.grad_views), by_entity):
            if sparse and v.dim() == 2:
                buf = st["bufs"].get(tuple(v.shape))
                if buf is None:
                    buf = st["bufs"][tuple(v.shape)] = torch.zeros(cap, v.shape[1], dtype=torch.float32, device=v.device)
                buf.zero_()
                buf[s_local] = g[ids]                    # (a row named twice is written twice with the same values)
                dist.all_reduce(buf, group=self.process_group)
                if mean:
                    buf.div_(self.world_size)
                g[ids_all] = buf[s_all]                  # every touched row, also those only other ranks touched
            else:                                        # relation-indexed tables: small, dense
                dist.all_reduce(g, group=self.process_group)
                if mean:
                    g.div_(self.world_size)

    def _collectives(self):
        """(reduce_scatter_ok, backend name) of the process group: RCCL ("nccl") has the fused primitives; gloo (CPU
        tests, ranks sharing one GPU) gets the same result from all_reduce + slice."""
        name = torch.distributed.get_backend(self.process_group)
        return name == "nccl", name

    def _reduce_and_step(self, advance=None, clear_local_grad=True, overlap_gather=False):
        """Gradient exchange (N > 1) + dense optimiser.  `advance`: the device-resident step-state arguments of
        FlatState.optimizer_step_advance when the step is being captured into a hipGraph."""
        flat = self.flat

        def optimise():
            if self._rescal_fused():
                # RESCAL, single GPU: the optimiser stores the entity rows already renormalised for the next step's forward
                # (kge_optimizer_step_rows) and the relation matrices are renormalised right behind it, so the next step skips its
                # normalisation pass -- except after the epoch's last step, which leaves the tables as the optimiser wrote
                # them (that is what the reference's tables hold when an epoch ends).
                keep = not self._rescal_last
                ent = flat.views[0]
                par, self._touched_step = self._touched_step, None
                bm = self._touched_bitmaps() if par is not None else (None, None)
                st, self._stage_step = getattr(self, "_stage_step", None), None
                rel = flat.views[1]
                rownorm = None
                if (keep and self.K is K and len(flat.views) == 2 and flat.offsets[1] + rel.numel() == flat.numel
                        and K.optimizer_step_rownorm_ok(rel.shape[0], rel.shape[1])):
                    rownorm = (rel.shape[0], rel.shape[1])
                done = flat.optimizer_step_rows_first(self.config.learning_rate, ent.shape[0], ent.shape[1], keep, advance,
                                                      touched=bm[par] if par is not None else None,
                                                      touched_clear=bm[1 - par] if par is not None else None,
                                                      stage=st if par is not None else None, rest_rownorm=rownorm,
                                                      rider=self.switches.get("opt_rider") is not False)
                self._touch_parity ^= 1
                if keep and not done:
                    self.K.rescal_normalize_relations(flat.views[1], self.model.hidden_size)
                self._rescal_normalised = keep
                return
            self._rescal_normalised = False
            if advance is not None:
                flat.optimizer_step_advance(self.config.learning_rate, *advance)
            else:
                flat.optimizer_step(self.config.learning_rate)

        if not self.distributed:
            optimise()
            return
        # Sharded data-parallel step: reduce-scatter the flat gradient (each rank receives the SUM -- or, for the
        # mean-type losses, the AVERAGE, since each rank already divided by its local row count -- of its 1/N shard),
        # run the dense optimiser on that shard only, all-gather the updated parameters in place.  Replicas hold
        # byte-identical tables afterwards by construction (everyone receives the same shards).
        dist = torch.distributed
        mean = self._mean_type_loss()
        if getattr(self, "_sparse_dp", False):
            # sparse exchange: gradient rows of the touched entities + the small dense tables, then the full optimiser on every rank
            self._mark("compute")
            self._sparse_exchange(mean)
            self._mark("reduce_scatter")
            optimise()
            self._mark("optimiser")
            return
        fused, _ = self._collectives()
        if getattr(self, "_dp_allreduce", False):
            # small tables: one all-reduce, every rank steps every row (replicas identical: all ranks hold the same reduced buffer)
            self._mark("compute")
            if fused:
                dist.all_reduce(flat.grad, op=dist.ReduceOp.AVG if mean else dist.ReduceOp.SUM, group=self.process_group)
            else:
                dist.all_reduce(flat.grad, group=self.process_group)
                if mean:
                    flat.grad.div_(self.world_size)
            self._mark("reduce_scatter")
            optimise()
            self._mark("optimiser")
            return
        self._wait_gather()
        self._mark("compute")
        if fused:
            dist.reduce_scatter_tensor(flat.grad_shard, flat.grad, op=dist.ReduceOp.AVG if mean else dist.ReduceOp.SUM,
                                       group=self.process_group)
        else:
            dist.all_reduce(flat.grad, group=self.process_group)
            flat.grad_shard.copy_(flat.grad[flat.shard_lo:flat.shard_lo + flat.shard_numel])
            if mean:
                flat.grad_shard.div_(self.world_size)
        if clear_local_grad:
            flat.grad.zero_()      # the local accumulation buffer of the next step (the optimiser clears only grad_shard)
        self._mark("reduce_scatter")
        optimise()
        self._mark("optimiser")
        if overlap_gather and advance is None and self.phase_marks is None:
            # eager steps inside an epoch: the all-gather runs on RCCL's stream while the host marshals -- and the GPU runs -- the
            # next batch's sampler; the first launch that reads the tables waits for it (_wait_gather).  Same values, same order.
            self._gather_work = dist.all_gather_into_tensor(flat.param, flat.param_shard, group=self.process_group, async_op=True)
        else:
            dist.all_gather_into_tensor(flat.param, flat.param_shard, group=self.process_group)
            self._mark("all_gather")

    def train_step_pairwise(self, pos_h, pos_r, pos_t, neg_h, neg_r, neg_t):
        """Loss of one batch as a device scalar (no sync); gradients are left in the flat buffer."""
        self.loss_buf.zero_()
        self._accumulate_pairwise(pos_h, pos_r, pos_t, neg_h, neg_r, neg_t)
        return self.K.read_loss(self.loss_buf)

    def train_step_pointwise(self, h, r, t, target):
        self.loss_buf.zero_()
        self._accumulate_pointwise(h, r, t, target)
        return self.K.read_loss(self.loss_buf)

    # ------------------------------------------------------------------ hipGraph replay of the whole step
    def _graph_wanted(self, num_batch=None):
        # captured steps read perm[batch_idx * B + i] for all i < B with no per-step clamp (the eager path clamps in
        # Generator._next_range): every batch of the epoch must lie inside the train permutation
        if num_batch is not None and num_batch * int(self.config.batch_size) > self.generator.n_train:
            return False
        if self.K is not K:
            return False
        if self.switches["staged"] and self._staged_ok():   # the staged step is an eager two-launch step
            return False
        if self.use_graph is None and self.generator is not None and self.step_path(1) in ("pull", "own", "transx"):
            return False   # one native call per epoch beats a replay per step
        if self.distributed:
            # RCCL collectives are capturable (gloo is not); multi-rank capture is opt-in (use_graph=True or
            # KGE_GRAPH_MULTI=1): it has only ever run on one-rank process groups (tests/test_hip_dist.py)
            asked = bool(self.use_graph) or (self.use_graph is None and bool(self.switches["graph_multi"]))
            return asked and self._collectives()[0] and self.config.batch_size % self.world_size == 0
        if self.use_graph is not None:
            return bool(self.use_graph)
        rows = self.config.batch_size * (1 + self.config.neg_rate)
        return rows <= self.GRAPH_MAX_ROWS

    def _capture_step(self, num_batch):
        """Capture [sample ->] fused step -> optimiser (+ next step's state) once per state parity; per-step values
        that change (batch position, Philox offset, Adam bias terms) live in device memory, so replays need no host
        arguments."""
        gen, cfg = self.generator, self.config
        dev = self.flat.param.device
        pointwise = self.model.training_strategy != TrainingStrategy.PAIRWISE_BASED
        B = int(cfg.batch_size)
        # two sets of device-resident step state {cursor[8], hyper[4]}: the step that reads set p leaves the state of the
        # following step in set 1-p (written by its optimiser launch), so a step is [sample +] fused step + optimiser
        # with no separate advance launch; two graphs, one per parity, are replayed alternately
        self._cursor = torch.zeros(16, dtype=torch.int64, device=dev)
        self._hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        cur = [self._cursor[0:8], self._cursor[8:16]]
        hyp = [self._hyper[0:4], self._hyper[4:8]]
        cur[0][2] = self.flat.step
        cur[0][4] = gen._draws
        self._sbuf = K.sample_buffer(B // self.world_size, gen.neg_rate, pointwise, dev)
        self._graph_batches = num_batch
        K.step_advance(cur[0], hyp[0], B, num_batch, B * gen.neg_rate, cfg.learning_rate)  # state of the first step

        per = B // self.world_size           # this rank's slice of every batch (Philox counters stay global)
        shard = (self.rank * per, per, self.rank * per * gen.neg_rate)

        def body(p):
            self._touch_parity = p
            self._accumulate_next_batch(cursor=cur[p], fixed_range=shard)
            self._reduce_and_step(advance=(hyp[p], cur[p], cur[1 - p], hyp[1 - p], B, num_batch, B * gen.neg_rate))

        self._graph_body = body
        body(0)  # the epoch's FIRST step runs eagerly (loads kernels, sizes workspaces) ...
        torch.cuda.synchronize()
        self._graphs = []
        for p in (0, 1):  # ... then the launch sequence of each parity is captured (capture does not execute)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body(p)
            self._graphs.append(graph)
        self._graph = self._graphs[0]
        self._parity = 1  # the eager step consumed set 0; the next step's state is in set 1
        # launch-bound steps: one graph launch per step still costs the host ~10 us; a third graph replays GRAPH_UNROLL
        # (even) steps back to back starting from parity 1
        self._graph_multi = None
        if self.GRAPH_UNROLL >= 2 and num_batch >= 2 * self.GRAPH_UNROLL:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for u in range(self.GRAPH_UNROLL):
                    body((1 + u) & 1)
            self._graph_multi = graph
        return 1  # steps already executed

    # ------------------------------------------------------------------ epochs
    def train_model_epoch(self, epoch_idx, tuning=False):
        num_batch = self.config.tot_train_triples // self.config.batch_size if not self.config.debug else 10
        self.generator.start_one_epoch(num_batch)
        self.model.train()
        # RESCAL (see _reduce_and_step): inside the epoch the optimiser hands the tables to the next step already renormalised
        self._in_epoch, self._rescal_normalised, self._rescal_last = True, False, num_batch == 1
        try:
            return self._train_epoch_body(epoch_idx, num_batch)
        except BaseException:
            # a step that died between the pair step's registrations and the optimiser that consumes (and resets) them would leave
            # entries in the per-entity lists of the staged RESCAL gradients: start the next epoch from empty lists
            # (after a HIP fault these device ops raise themselves: the ORIGINAL error is the one to surface)
            self._touched_step, self._stage_step = None, None
            st = getattr(self, "_rescal_stage", None)
            if st is not None:
                try:
                    st.count.zero_(); st.head.zero_()
                    for b in (self._touched or ()):
                        b.zero_()
                except Exception as cleanup_error:   # noqa: BLE001
                    _log("train_model_epoch: resetting the staged-gradient lists after a failed step also failed: %r" % (cleanup_error,))
            raise
        finally:
            self._in_epoch, self._rescal_normalised, self._rescal_last = False, False, False

    def _train_epoch_body(self, epoch_idx, num_batch):
        if num_batch > 0 and self._graph_wanted(num_batch):
            self.loss_buf.zero_()
            gen = self.generator
            step0, draws0 = self.flat.step, gen._draws  # host mirrors of the device-resident counters
            done = 0
            fused = self._rescal_fused()
            if self._graph is None or self._graph_batches != num_batch:
                done = self._capture_step(num_batch)
            elif fused:   # the captured steps expect renormalised tables (their normalisation rides in the previous optimiser launch)
                self.K.rescal_normalize(self.flat.views[0], self.flat.views[1], self.model.hidden_size)
                self._rescal_normalised = True
            # (an epoch is exactly num_batch steps, so the device-side batch index has wrapped to 0 by itself: every
            # epoch walks the permutation from its start, data/generator.py:28-35)
            remaining = num_batch - done
            tail = 1 if (fused and remaining > 0) else 0   # RESCAL: the epoch's last step runs eagerly, without the renormalisation
            while remaining > tail:
                if self._graph_multi is not None and self._parity == 1 and remaining - tail >= self.GRAPH_UNROLL:
                    self._graph_multi.replay()  # an even number of steps: parity unchanged
                    remaining -= self.GRAPH_UNROLL
                else:
                    self._graphs[self._parity].replay()
                    self._parity ^= 1
                    remaining -= 1
            if tail:
                self._rescal_last = True
                self._graph_body(self._parity)
                self._parity ^= 1
            self.flat.step = step0 + num_batch
            gen._draws = draws0 + num_batch * gen.batch_size * gen.neg_rate
            gen._pending = 0
        else:
            self.loss_buf.zero_()
            if self._pull_ok():
                self._pull_state()[0].sync_in()   # the tables may have been changed from outside since the last epoch
            self.step_next_batches(num_batch)
            self.sync_model()
        acc = self.K.read_loss(self.loss_buf)
        if self.distributed:
            torch.distributed.all_reduce(acc, group=self.process_group)
            if self._mean_type_loss():
                acc = acc / self.world_size  # average of per-rank means
        acc_loss = float(acc.item())  # the only host sync of the epoch
        self.training_results.append([epoch_idx, acc_loss])
        return acc_loss

    def _new_generator(self):
        gen = Generator(self.model, self.config, rank=self.rank, world_size=self.world_size, backend=self.K)
        if self.K is K and not self.distributed and self.model.kernel_name in ("transe", "transm") and self._pull_two_phase():
            # two-phase step: a visit is a few bytes, so an item carries as many incidences as its lane group is wide (32; 16 with KGE_PULL_G=16)
            gen.pull_segment = min(32, 256 // max(1, self.K.pull_groups_per_block(self.model.hidden_size)))
        return gen

    # ------------------------------------------------------------------ checkpoints (reference format, utils/trainer.py:388-419)
    def _checkpoint_dir(self, path=None):
        if path is not None:
            return str(path)
        root = getattr(self.config, "path_tmp", None)
        if root is None:
            raise ValueError("save_model / load_model: no path given and the configuration has no path_tmp")
        return os.path.join(str(root), self.model.model_name)

    def save_model(self, path=None):
        """state_dict under the reference's file name and key names (`ent_embeddings.weight`, ...) plus the pickled configuration
        next to it (`config.npy`): the pair the reference's load_model requires.  Default directory as the reference's:
        config.path_tmp / model_name."""
        import numpy as np
        self.sync_model()
        d = self._checkpoint_dir(path)
        os.makedirs(d, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(d, self.TRAINED_MODEL_FILE_NAME))
        # the configuration is pickled to a temporary file and renamed on success: a configuration holding an unpicklable object (open
        # handles, lambdas) must not leave a truncated config.npy next to the weights -- the reference's load_model needs both files
        # (utils/trainer.py:408) and would fail on a partial one with an unrelated error
        final = os.path.join(d, self.TRAINED_MODEL_CONFIG_NAME)
        tmp = final + ".tmp%d" % os.getpid()
        try:
            with open(tmp, "wb") as f:
                np.save(f, self.config, allow_pickle=True)
            os.replace(tmp, final)
        except Exception as e:
            if os.path.exists(tmp):
                os.remove(tmp)
            if os.path.exists(final):   # a stale configuration from an earlier save no longer describes these weights
                os.remove(final)
            _log("save_model: configuration not pickled, checkpoint is weights only (%s: %s)" % (type(e).__name__, e))

    def load_model(self, path=None):
        """Load a checkpoint written by save_model or by the reference.  Strict: every key the model's state_dict has must be in
        the file with the same shape, and the file may hold nothing else -- a wrong or partial checkpoint raises instead of loading
        silently.  (The reference additionally swaps in the pickled configuration; the tables' shapes are what this path checks.)"""
        d = self._checkpoint_dir(path)
        f = os.path.join(d, self.TRAINED_MODEL_FILE_NAME)
        if not os.path.exists(f):
            raise ValueError("Cannot load model from %s" % d)   # utils/trainer.py:418-419
        state = torch.load(f, map_location=self.config.device)
        own = self.model.state_dict()
        missing, extra = sorted(set(own) - set(state)), sorted(set(state) - set(own))
        wrong = sorted(k for k in set(own) & set(state) if tuple(own[k].shape) != tuple(state[k].shape))
        if missing or extra or wrong:
            raise ValueError("load_model: checkpoint does not match the model (missing %s, unexpected %s, shape mismatch %s)"
                             % (missing, extra, [(k, tuple(state[k].shape), tuple(own[k].shape)) for k in wrong]))
        if self.flat is None:
            self.model.load_state_dict(state, strict=True)
            return
        with torch.no_grad():   # in place: the parameters live in the flat buffer
            for k, v in own.items():
                v.copy_(state[k])
        if getattr(self, "_pull", None) is not None:
            self._pull.sync_in()

    def _is_better(self, metrics):
        key = self.monitor.value
        if self.monitor in (Monitor.MEAN_RANK, Monitor.FILTERED_MEAN_RANK):
            return metrics[key] < self.best_metric[key]
        return metrics[key] > self.best_metric[key]

    def train_model(self):
        self.generator = self._new_generator()
        cur_epoch_idx = 0
        for cur_epoch_idx in range(self.config.epochs):
            _log("Epoch[%d/%d]" % (cur_epoch_idx, self.config.epochs))
            loss = self.train_model_epoch(cur_epoch_idx)
            _log("acc_loss: %f" % loss)
            if cur_epoch_idx % self.config.test_step == 0:
                self.model.eval()
                with torch.no_grad():
                    metrics = self.evaluator.mini_test(cur_epoch_idx)
                if self.early_stopper is not None and self.early_stopper.should_stop(metrics):
                    break
                if getattr(self.config, "save_model", False) and getattr(self.config, "path_tmp", None) is not None:
                    # keep the best weights seen so far (utils/trainer.py:207-219)
                    if self.best_metric is None or self._is_better(metrics):
                        self.best_metric = metrics
                        # replicas are identical: one writer (N truncating writers would race on the same files).  The barrier is reached
                        # whatever the save does: a rank-0 exception must not leave the other ranks waiting in it
                        try:
                            if self.rank == 0:
                                self.save_model()
                        finally:
                            if self.distributed:
                                torch.distributed.barrier(group=self.process_group)
        self.model.eval()
        with torch.no_grad():
            self.evaluator.full_test(cur_epoch_idx)
        self.generator.stop()
        return cur_epoch_idx
