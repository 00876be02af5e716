"""State holders of the drop-in Trainer (pykg2vec_amd/trainer.py): the patience rule of the reference's early stopping
(utils/trainer.py:22-68), the flat parameter / gradient / optimiser-state buffers every step path works on, and the double-buffered
tables of the owner-computes TransE step.  Split out of trainer.py; `pykg2vec_amd.trainer` re-exports all three."""
import torch

from . import kernels as K
from .common import Monitor


def _log(msg):
    print(msg, flush=True)


class PatienceStopper:
    """The early-stopping rule `train_model` applies after every mini-test (utils/trainer.py:22-68, 199-205): compare the monitored
    metric with the PREVIOUS mini-test's; a worse value spends one unit of patience, or stops the training when none is left; any
    other value refills the patience.  Lower is better for the (filtered) mean rank, higher for the reciprocal ranks.  A negative
    patience never stops (there is never exactly zero left)."""

    def __init__(self, patience, monitor):
        self.patience, self.monitor = int(patience), monitor
        self.left, self.previous = int(patience), None

    def should_stop(self, metrics):
        cur = metrics[self.monitor.value]
        prev, self.previous = self.previous, cur
        if prev is None:
            return False
        worse = cur > prev if self.monitor in (Monitor.MEAN_RANK, Monitor.FILTERED_MEAN_RANK) else cur < prev
        if worse and self.left > 0:
            self.left -= 1
            _log("%d more chances before the trainer stops the training. (prev_%s, curr_%s): (%.4f, %.4f)"
                 % (self.left, self.monitor.name, self.monitor.name, prev, cur))
            return False
        if worse and self.left == 0:
            _log("Stop the training.")
            return True
        self.left = self.patience
        return False


class FlatState:
    """All parameter tables of a model re-homed into one flat fp32 buffer (16-byte aligned segments) with matching
    flat gradient and optimiser-state buffers: one optimiser launch per step and one pair of collectives.

    Data parallel (world_size N > 1): the flat buffers are padded to a multiple of 4*N floats and cut into N equal
    shards.  Every rank keeps the full parameters and its full local gradient, but the optimiser STATE (Adam moments,
    Adagrad / RMSprop accumulators) only for its own shard, and runs the dense optimiser sweep only over that shard:
    reduce-scatter(grad) -> optimiser on 1/N of the tables -> all-gather(param).  Same bytes on the wire as an
    all-reduce, 1/N of the optimiser sweep (the B-independent part of the step) and of its state memory."""

    def __init__(self, model, optimizer, backend=K, world_size=1, rank=0, distributed=None, replicate_optimizer=False):
        """replicate_optimizer (data parallel with the sparse gradient exchange, Trainer._sparse_dp): every rank keeps the optimiser
        state of ALL rows and steps all of them, so no parameter all-gather exists -- the shard is the whole buffer."""
        self.K = backend
        # distributed (default: world_size > 1): the step goes through the collectives and needs a reduce-scatter target that
        # is distinct from the local gradient buffer -- also at world size 1 when a process group was given explicitly
        distributed = world_size > 1 if distributed is None else bool(distributed)
        params = [p.weight for p in model.parameter_list]
        dev = params[0].device
        offs, tot = [], 0
        for p in params:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        self.offsets = offs   # float offset of every table in the flat buffers
        quantum = 4 * world_size
        tot = (tot + quantum - 1) // quantum * quantum
        self.numel = tot
        self.world_size, self.rank = world_size, rank
        self.replicate_optimizer = bool(replicate_optimizer)
        self.shard_numel = tot if replicate_optimizer else tot // world_size
        self.shard_lo = 0 if replicate_optimizer else rank * self.shard_numel
        self.param = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.views, self.grad_views = [], []
        for p, o in zip(params, offs):
            v = self.param[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v  # nn.Parameter keeps its name/shape; storage is now the flat buffer
            self.views.append(v)
            self.grad_views.append(self.grad[o:o + p.numel()].view_as(p))
        self.optimizer = optimizer
        self.param_shard = self.param[self.shard_lo:self.shard_lo + self.shard_numel]
        # reduced gradient of this rank's shard: the full buffer itself when there is nothing to reduce
        self.grad_shard = self.grad if (not distributed or replicate_optimizer) else torch.zeros(self.shard_numel, dtype=torch.float32, device=dev)
        self.state1 = torch.zeros_like(self.param_shard) if optimizer in ("adam", "adagrad", "rms") else None
        self.state2 = torch.zeros_like(self.param_shard) if optimizer == "adam" else None
        self.step = 0

    def optimizer_step_advance(self, lr, hyper, cursor, next_cursor, next_hyper, batch_stride, n_batches, draws):
        self.step += 1
        self.K.optimizer_step_advance(self.optimizer, self.param_shard, self.grad_shard, self.state1, self.state2, lr, hyper,
                                      cursor, next_cursor, next_hyper, batch_stride, n_batches, draws, zero_grad=True)

    def optimizer_step_rows_first(self, lr, rows, dim, normalize, advance=None, touched=None, touched_clear=None, stage=None, rest_rownorm=None,
                                  rider=True):
        """The optimiser step with the FIRST table ([rows, dim], at offset 0 of the flat buffers) handled by the row-owner kernel
        (which can store the rows renormalised: RESCAL) and the remaining tables by the flat sweep.  advance: as
        optimizer_step_advance (device-resident step state of hipGraph-replayed steps); single GPU only."""
        self.step += 1
        n0 = rows * dim
        cut = self.offsets[1] if len(self.offsets) > 1 else self.numel
        sl = lambda buf, a, b: buf[a:b] if buf is not None else None
        hyper = advance[0] if advance is not None else None
        if (rest_rownorm is not None and normalize and rider and hasattr(self.K, "optimizer_step_rows_rownorm")
                and dim % 4 == 0 and dim <= 1024):     # (the rider form exists for rows of float4s only: k_opt_rows4)
            # RESCAL: the relation matrices' optimiser rides in the first workgroups of the entity table's sweep, one rescale launch
            # follows (kge_optimizer_step_rows_rownorm, round 6: two launches instead of three, bit-identical tables)
            r_rows, r_dim = rest_rownorm
            n1 = r_rows * r_dim
            self.K.optimizer_step_rows_rownorm(self.optimizer, self.param[:n0], self.grad[:n0], sl(self.state1, 0, n0), sl(self.state2, 0, n0),
                                               rows, dim, self.param[cut:cut + n1], self.grad[cut:cut + n1], sl(self.state1, cut, cut + n1),
                                               sl(self.state2, cut, cut + n1), r_rows, r_dim, lr, self.step, normalize=True, dev_hyper=hyper,
                                               touched=touched, touched_clear=touched_clear, stage=stage, advance=advance)
            return True
        if stage is not None:   # entity gradients staged by the pair step (kernels.RescalStage): summed per row in a fixed order, no atomics
            self.K.optimizer_step_rows_staged(self.optimizer, self.param[:n0], self.grad[:n0], sl(self.state1, 0, n0), sl(self.state2, 0, n0),
                                              rows, dim, lr, self.step, stage, touched, touched_clear, normalize=normalize, dev_hyper=hyper)
        else:
            self.K.optimizer_step_rows(self.optimizer, self.param[:n0], self.grad[:n0], sl(self.state1, 0, n0), sl(self.state2, 0, n0),
                                       rows, dim, lr, self.step, zero_grad=True, normalize=normalize, dev_hyper=hyper,
                                       touched=touched, touched_clear=touched_clear)
        rest = (self.param[cut:], self.grad[cut:], sl(self.state1, cut, self.numel), sl(self.state2, cut, self.numel))
        if rest_rownorm is not None:
            # the remaining table is [rows, dim] of wide rows to be renormalised behind its sweep (RESCAL's relation matrices): the
            # optimiser launch leaves the rows' sums of squares, one rescale launch follows (kge_optimizer_step_rownorm)
            r_rows, r_dim = rest_rownorm
            n1 = r_rows * r_dim
            self.K.optimizer_step_rownorm(self.optimizer, rest[0][:n1], rest[1][:n1], sl(rest[2], 0, n1), sl(rest[3], 0, n1), r_rows, r_dim,
                                          lr, self.step, zero_grad=True, advance=advance)
            return True
        if advance is not None:
            self.K.optimizer_step_advance(self.optimizer, *rest, lr, *advance, zero_grad=True)
        else:
            self.K.optimizer_step(self.optimizer, *rest, lr, self.step, zero_grad=True)
        return False

    def optimizer_step(self, lr, dev_hyper=None):
        """Dense optimiser sweep over this rank's shard (the whole buffer when world_size == 1); clears the reduced
        gradient it consumed."""
        self.step += 1
        self.K.optimizer_step(self.optimizer, self.param_shard, self.grad_shard, self.state1, self.state2, lr, self.step,
                              zero_grad=True, dev_hyper=dev_hyper)


class PullState:
    """Device state of the owner-computes training step (csrc/kge_pull.hip): the second half of the double-buffered
    tables, the row norms of both halves and the per-step sampler lists.  `cur` = which half holds the current tables
    (0 = FlatState.param, i.e. the storage behind the model's nn.Parameters)."""

    def __init__(self, flat, model, batch_size, max_slots, grad_only=False, two_phase=False):
        dev = flat.param.device
        self.flat = flat
        # grad_only (data-parallel ranks): the step writes gradient rows into FlatState.grad instead of updated tables, so
        # there is no second half -- `tables[1]` are the gradient views and only hats[0] / norms[0] are used
        self.grad_only = grad_only
        self.alt = flat.grad if grad_only else torch.empty_like(flat.param)
        shapes = [tuple(v.shape) for v in flat.views[:2]]
        offs = [v.data_ptr() - flat.param.data_ptr() for v in flat.views[:2]]
        view = lambda buf: [buf[o // 4:o // 4 + r * d].view(r, d) for o, (r, d) in zip(offs, shapes)]
        self.tables = [view(flat.param), view(self.alt)]
        # row-normalised copies of both halves, rows of kge_pull_hat_stride floats (compact since round 6: 6.5 instead of 8.3 MB at C1)
        stride = K.pull_hat_stride(shapes[0][1])
        # (entity rows, then relation rows, in ONE buffer per half: a single kge_row_norms call refreshes both tables when they
        # are adjacent in the flat parameter buffer)
        self.hat_all = [torch.zeros(sum(r for r, _ in shapes), stride, dtype=torch.float32, device=dev) for _ in range(2)]
        self.hats = [[h[:shapes[0][0]], h[shapes[0][0]:]] for h in self.hat_all]
        self.state1 = view(flat.state1) if flat.state1 is not None and not grad_only else None
        self.state2 = view(flat.state2) if flat.state2 is not None and not grad_only else None
        E, R, d = shapes[0][0], shapes[1][0], shapes[0][1]
        self.E, self.R = E, R
        self.norms = [torch.empty(E + R, dtype=torch.float32, device=dev) for _ in range(2)]
        self.batch_size = batch_size
        # two sampler list sets: the step on batch k consumes one while the sampler of batch k+1, riding in the same
        # launch, fills the other.  `ready` = (batch index, Philox offset) the current set was sampled for.
        self.lists = [K.PullListSet(batch_size, E, dev) for _ in range(2)]
        self.cur_list = 0
        self.ready = None
        self.calls = {}   # prepared kge_pull_step calls of the data-parallel gradient step
        self.partials = torch.empty(max(1, max_slots) * K.pull_partial_stride(d), dtype=torch.float32, device=dev)
        # two-phase ("staged direction") form: every pair evaluated once (k_pull_eval), the owners sum its record
        self.direction = K.PullDirection(batch_size, d, bool(getattr(model, "l1_flag", True)), dev) if two_phase else None
        self.cur = 0

    def sync_in(self):
        """(Re)derive the row norms from the tables the model currently holds (they may have been set from outside)."""
        if self.grad_only:
            self.ready = None
            self.refresh_norms()
            return
        self.sync_out()
        K.row_norms(self.tables[0][0], self.norms[0][:self.E], self.hats[0][0])
        K.row_norms(self.tables[0][1], self.norms[0][self.E:], self.hats[0][1])

    def refresh_norms(self):
        """Row norms / normalised copies of the current parameter tables (after the all-gather of a data-parallel step)."""
        ent, rel = self.tables[0]
        if rel.data_ptr() == ent.data_ptr() + ent.numel() * 4 and ent.shape[1] == rel.shape[1]:   # adjacent: one launch
            both = self.flat.param[(ent.data_ptr() - self.flat.param.data_ptr()) // 4:][:ent.numel() + rel.numel()]
            K.row_norms(both.view(self.E + self.R, ent.shape[1]), self.norms[0], self.hat_all[0])
            return
        K.row_norms(ent, self.norms[0][:self.E], self.hats[0][0])
        K.row_norms(rel, self.norms[0][self.E:], self.hats[0][1])

    def sync_out(self):
        """Make FlatState.param (the storage behind the model's parameters) hold the current tables."""
        if self.cur == 1 and not self.grad_only:
            self.flat.param.copy_(self.alt)
            self.norms[0].copy_(self.norms[1])
            for t in (0, 1):
                self.hats[0][t].copy_(self.hats[1][t])
            self.cur = 0
