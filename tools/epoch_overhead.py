"""Dev tool: where an epoch of the owner-computes path spends its time beyond its steps (C1, B = 32 768: 14 steps of 27 us per epoch):
host-side phase timers around Trainer.train_model_epoch's parts, with and without the per-epoch host read of the loss.  One MI355X."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench

dev = torch.device("cuda:0")
H = bench.setup_headline(32768, 64, dev)
tr, gen = H.tr, H.gen
nb = H.steps_per_epoch
for _ in range(5):
    tr.train_model_epoch(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
E_ = 50
for e in range(E_):
    tr.train_model_epoch(e)
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / E_ * 1e6
print("train_model_epoch: %.1f us per epoch of %d steps = %.1f us per step" % (whole, nb, whole / nb))
# the same epochs, phase by phase (host timers; a synchronize after each phase so that the phase's GPU work is inside it)
acc = {}
def T(name, fn):
    t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t); return r
for e in range(E_):
    T("start_one_epoch + flags", lambda: (gen.start_one_epoch(nb), tr.model.train()))
    T("loss_buf.zero_", lambda: tr.loss_buf.zero_())
    T("_pull_ok", lambda: tr._pull_ok())
    T("sync_in", lambda: tr._pull_state()[0].sync_in())
    T("step_next_batches", lambda: tr.step_next_batches(nb))
    T("sync_model", lambda: tr.sync_model())
    T("read_loss + item", lambda: float(tr.K.read_loss(tr.loss_buf).item()))
tot = sum(acc.values()) / E_ * 1e6
for k, v in acc.items():
    print("   %-28s %8.1f us" % (k, v / E_ * 1e6))
print("   sum %.1f us (each phase synchronised: serial upper bound)" % tot)
# steps only, back to back, no per-epoch work
torch.cuda.synchronize()
t0 = time.perf_counter()
for e in range(E_):
    gen.start_one_epoch(nb)
    tr.step_next_batches(nb)
torch.cuda.synchronize()
print("steps only: %.1f us per epoch = %.2f us per step" % ((time.perf_counter() - t0) / E_ * 1e6, (time.perf_counter() - t0) / E_ / nb * 1e6))
