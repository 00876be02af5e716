#!/usr/bin/env python
"""Feasibility probe for overlapping C4's memory-bound entity-table optimiser sweep (k_opt_rows4, ~110 us) with the latency-bound
pair kernel (k_rescal_pair, ~47 us) on two HIP streams: wall time of [sweep ; pair] back to back on one stream against the two on
separate streams (high priority for the pair kernel), same inputs, results irrelevant.  Usage: overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tools import bench_extra
from pykg2vec_amd import kernels as K

dev = "cuda:0"
c, cfg, model, tr, q, steps = bench_extra.build_extra_config("C4", dev, steps_cap=20)
tr.train_model_epoch(0)
flat, ent = tr.flat, tr.flat.views[0]
n0, E, k = ent.numel(), ent.shape[0], ent.shape[1]
B = cfg.batch_size
rng = np.random.default_rng(0)
mk = lambda hi: torch.from_numpy(rng.integers(hi, size=B)).to(dev)
ph, pt, nh, nt, pr = mk(E), mk(E), mk(E), mk(E), torch.sort(mk(cfg.tot_relation)).values
bm = tr._touched_bitmaps()
side = torch.cuda.Stream(priority=0)
hi = torch.cuda.Stream(priority=-1)

def sweep(t):
    K.optimizer_step_rows("adam", flat.param[:n0], flat.grad[:n0], flat.state1[:n0], flat.state2[:n0], E, k, 0.01, t, normalize=True,
                          touched=bm[0], touched_clear=bm[1])
def pair():
    K.rescal_pair_step(tr._desc, ph, pr, pt, nh, nt, 1.0, tr.loss_buf, touched=bm[0])

def timed(fn, reps=20):
    for _ in range(3): fn(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps): fn(i + 2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

def serial(t):
    sweep(t); pair()
def overlapped(t):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur); hi.wait_stream(cur)
    with torch.cuda.stream(side): sweep(t)
    with torch.cuda.stream(hi): pair()
    cur.wait_stream(side); cur.wait_stream(hi)
print("sweep alone   %.1f us" % timed(lambda t: sweep(t)))
print("pair alone    %.1f us" % timed(lambda t: pair()))
print("serial        %.1f us" % timed(serial))
print("two streams   %.1f us" % timed(overlapped))

# the same inside hipGraphs (no host work between launches): 8 repetitions per graph
def graph_of(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(8):
            fn(i + 2)
    return g
for name, fn in (("serial", serial), ("two streams", overlapped)):
    g = graph_of(fn)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print("graph, %-12s %.1f us per (sweep + pair)" % (name, (time.perf_counter() - t0) / 80 * 1e6))
