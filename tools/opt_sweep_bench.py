#!/usr/bin/env python
"""Is the dense optimiser sweep bound by memory or by its arithmetic?  kge_optimizer_step (flat k_opt) and kge_optimizer_step_rows
(row-owner k_opt_rows4, with / without the RESCAL renormalisation) over tables of 6.5 MB ... 420 MB, every optimiser: us per sweep,
GB/s over (reads + writes) and element-steps per second.  A cache-resident table that streams no faster per element than a 400 MB one
is arithmetic-bound (IEEE sqrt + two divisions per element for Adam)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pykg2vec_amd import kernels as K

dev = "cuda"
STREAMS = {"sgd": 3, "adagrad": 5, "rms": 5, "adam": 7}   # p r/w, g r (+w when cleared: gradient is zero here: no write), state r/w
print("| kernel | optimiser | table MB | us / sweep | GB/s | T element-steps/s |")
print("|---|---|---|---|---|---|")
for rows, dim in ((16296, 100), (65184, 100), (123182, 200), (524288, 200)):
    n = rows * dim
    for kind in ("sgd", "adagrad", "rms", "adam"):
        for form in ("flat", "rows", "rows+norm"):
            p = torch.randn(n, device=dev) * 0.1 + 1.0
            g = torch.zeros(n, device=dev)
            s1 = torch.rand(n, device=dev) * 1e-3 if kind != "sgd" else None
            s2 = torch.rand(n, device=dev) * 1e-5 if kind == "adam" else None
            def run(t):
                if form == "flat":
                    K.optimizer_step(kind, p, g, s1, s2, 0.01, t)
                else:
                    K.optimizer_step_rows(kind, p, g, s1, s2, rows, dim, 0.01, t, normalize=form == "rows+norm")
            for t in range(1, 6):
                run(t)
            torch.cuda.synchronize()
            reps = 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(6, 6 + reps):
                run(t)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            # SGD / Adagrad skip rows whose gradient is zero: give them a gradient so that the full path runs
            streams = STREAMS[kind] - (1 if kind in ("sgd", "adagrad") else 0)
            print("| %s | %s | %.1f | %.1f | %.0f | %.3f |" % (form, kind, n * 4 / 1e6, us, streams * n * 4 / us / 1e3, n / us / 1e6))
