"""Dev tool: the relation-matrix models (RESCAL, TransR, NTN -- f32 MFMA paths) at B in {128, 4096, 32768}: time of the fused
scoring + loss + backward step on an explicit batch (no sampler, no optimiser) and the matrix-core rate it implies, to
separate "launch-sized batch" from kernel quality.  Prints a markdown table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_util
from pykg2vec_amd import kernels as K
from pykg2vec_amd.trainer import Trainer

CASES = [  # name, model, E, R, hp, flop per scored triple (forward + backward matrix products)
    ("RESCAL YAGO3-10 shape k=200", "rescal", 123182, 37, dict(hidden_size=200, margin=1.0), lambda hp: 6 * hp["hidden_size"] ** 2),
    ("RESCAL FB15k shape k=50", "rescal", 14951, 1345, dict(hidden_size=50, margin=1.0), lambda hp: 6 * hp["hidden_size"] ** 2),
    ("TransR FB15k shape 50/50", "transr", 14951, 1345, dict(ent_hidden_size=50, rel_hidden_size=50, l1_flag=True, margin=1.0),
     lambda hp: 2 * 6 * hp["ent_hidden_size"] * hp["rel_hidden_size"]),
    ("TransR WN18RR shape 100/100", "transr", 40943, 11, dict(ent_hidden_size=100, rel_hidden_size=100, l1_flag=False, margin=1.0),
     lambda hp: 2 * 6 * hp["ent_hidden_size"] * hp["rel_hidden_size"]),
    ("NTN FB15k shape d=k=100", "ntn", 14951, 1345, dict(ent_hidden_size=100, rel_hidden_size=100, lmbda=1e-4, margin=1.0),
     lambda hp: 6 * hp["rel_hidden_size"] * hp["ent_hidden_size"] ** 2),
]
print("| model | B | step (fwd + loss + bwd) | TFLOP/s (matrix products) | fraction of 157 TF |")
print("|---|---|---|---|---|")
rng = np.random.default_rng(0)
for name, model, E, R, hp, flop in CASES:
    for B in (128, 4096, 32768):
        if model == "ntn" and B > 4096:
            continue   # 2 * k_r * d^2 flop per triple and a [B, k_r, d] intermediate: the reference's own presets use B = 128
        train = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
        cfg = hip_util.make_config(E, R, dict(hp, neg_rate=1), train[:64], train[:4], train[:4], optimizer="sgd", lr=0.01, batch_size=B)
        torch.manual_seed(0)
        m = hip_util.model_from_params(model, {}, hp, E, R, train=train)
        tr = Trainer(m, cfg, use_graph=False)
        tr.build_model()
        ph, pr, pt = [hip_util.dev(train[:, k]) for k in range(3)]
        nh = hip_util.dev(rng.integers(E, size=B)); nr = pr.clone(); nt = pt.clone()
        for _ in range(3):
            tr._accumulate_pairwise(ph, pr, pt, nh, nr, nt)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            tr._accumulate_pairwise(ph, pr, pt, nh, nr, nt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tf = 2 * B * flop(hp) / dt / 1e12
        print(f"| {name} | {B} | {dt * 1e6:.0f} us | {tf:.2f} | {tf / 157:.3f} |", flush=True)
        del tr, m
        torch.cuda.empty_cache()
