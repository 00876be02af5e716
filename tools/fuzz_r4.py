#!/usr/bin/env python
"""Randomised sweeps of the round-4 entry points on one MI355X:
  csr    kge_filter_csr_* against the dict-of-sets form (random E / R / split sizes / duplicates / absent keys)
  ids    debug mode: one bad id at a random position of a random id-taking entry point must be reported at exactly that position
Usage: ITERS=60 SEED=1 python tools/fuzz_r4.py      (exit code 1 on any violation)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pykg2vec_amd import kernels as K
from pykg2vec_amd._lib import KgeHipError

ITERS = int(os.environ.get("ITERS", "60"))
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
dev = "cuda"
mk = lambda a, dt=torch.int64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
bad = 0

# ---------------------------------------------------------------- csr
n_q = 0
for it in range(ITERS):
    E = int(rng.choice([3, 17, 200, 5000, 40943]))
    R = int(rng.choice([1, 2, 11, 237, 1345]))
    M = int(rng.integers(1, 40000))
    n = int(rng.integers(1, 3000))
    known = np.stack([rng.integers(E, size=M), rng.integers(R, size=M), rng.integers(E, size=M)], 1).astype(np.int64)
    if rng.random() < 0.5:
        known = np.concatenate([known, known[rng.integers(M, size=M // 2)]])
    q = known[rng.integers(len(known), size=n)].copy()
    miss = rng.random(n) < 0.2
    q[miss, 0] = rng.integers(E, size=int(miss.sum()))
    t_off, t_ids, h_off, h_ids = (x.cpu().numpy() for x in K.filter_csr_build(mk(known), mk(q), E, R))
    hr, tr = {}, {}
    for h, r, t in known.tolist():
        hr.setdefault((h, r), set()).add(t)
        tr.setdefault((t, r), set()).add(h)
    for i, (h, r, t) in enumerate(q.tolist()):
        if list(t_ids[t_off[i]:t_off[i + 1]]) != sorted(hr.get((h, r), ())) or list(h_ids[h_off[i]:h_off[i + 1]]) != sorted(tr.get((t, r), ())):
            bad += 1
            print("csr mismatch", it, i, E, R, M)
            break
    n_q += n
print("csr: %d cases, %d queries, bad so far %d" % (ITERS, n_q, bad), flush=True)

# ---------------------------------------------------------------- debug ids
K.set_debug(True)
try:
    for it in range(ITERS):
        E, R, d, n = int(rng.integers(5, 3000)), int(rng.integers(1, 50)), int(rng.choice([8, 20, 100])), int(rng.integers(1, 5000))
        ent, rel = torch.randn(E, d, device=dev), torch.randn(R, d, device=dev)
        desc = K.make_desc("transe", [ent, rel], None, tot_entity=E, tot_relation=R, dim=d, l1_flag=True)
        h, r, t = rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)
        K.score_forward(desc, mk(h), mk(r), mk(t))                       # clean: no error
        col, pos = int(rng.integers(3)), int(rng.integers(n))
        arrs = [h.copy(), r.copy(), t.copy()]
        arrs[col][pos] = [E, R, E][col] + int(rng.integers(0, 5)) if rng.random() < 0.7 else -int(rng.integers(1, 9))
        try:
            K.score_forward(desc, *[mk(a) for a in arrs])
            bad += 1
            print("ids: bad id not reported", it)
        except KgeHipError as e:
            if ("position %d:" % pos) not in str(e) or ["head", "relation", "tail"][col] not in str(e):
                bad += 1
                print("ids: wrong report", it, col, pos, e)
finally:
    K.set_debug(False)
print("ids: %d cases, bad so far %d" % (ITERS, bad), flush=True)
print("fuzz_r4: %s" % ("OK" if bad == 0 else "%d VIOLATIONS" % bad))
sys.exit(1 if bad else 0)
