"""Dev tool: what a timed region of K owner-computes steps costs beyond K x the steady-state step (bench.py's 20-step regions read
~2 us per step above its 200-step regions): wall clock between synchronize()s, the host's return from the enqueue, and the HIP-event
time between the first and the last launch, for K = 1 .. 14 (one epoch of the headline workload).  One MI355X."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench

dev = torch.device("cuda:0")
H = bench.setup_headline(32768, 64, dev)
bench.run_headline_steps(H, 64)
torch.cuda.synchronize()
for K in (1, 2, 4, 8, 14, 20, 28, 56):
    rows = []
    for rep in range(15):
        bench.reset_headline(H)
        H.gen._pending = 0           # every region starts at an epoch boundary (14 steps per epoch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        bench.run_headline_steps(H, K)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(((t2 - t0) * 1e6, (t1 - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
    w, h, e = (float(np.median([r[i] for r in rows])) for i in range(3))
    print("K=%3d  wall %8.1f us (%.2f per step)   host enqueue returns after %7.1f us   events %8.1f us (%.2f per step)" % (K, w, w / K, h, e, e / K), flush=True)
