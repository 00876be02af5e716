"""Dev tool: run only one rank sweep (for rocprofv3 counter passes).  SHAPE = transe (FB15k d=100, default) | c2 (ComplEx
WN18RR d=200, 3 134 triples) | c3 (RotatE FB15k-237 d=1000, 2 048) | c4 (RESCAL YAGO3-10 k=200, 1 024); N overrides the count."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_util
from pykg2vec_amd.evaluator import Evaluator
SHAPES = {"transe": ("transe", 14951, 1345, 8192, dict(hidden_size=100, l1_flag=bool(int(os.environ.get("L1", 1))), margin=1.0)),
          "c2": ("complex", 40943, 11, 3134, dict(hidden_size=200, lmbda=1e-4)),
          "c3": ("rotate", 14541, 237, 2048, dict(hidden_size=1000, margin=24.0, alpha=1.0)),
          "c4": ("rescal", 123182, 37, 1024, dict(hidden_size=200, margin=1.0))}
name, E, R, n, hp = SHAPES[os.environ.get("SHAPE", "transe")]
name = os.environ.get("MODEL", name)
n = int(os.environ.get("N", n))
rng = np.random.default_rng(1234)
test = np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1)
cfg = hip_util.make_config(E, R, hp, test[:10], test[:10], test)
m = hip_util.model_from_params(name, {}, hp, E, R)
ev = Evaluator(m, cfg)
for _ in range(3):
    ev.rank_all(test, n)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", 3))
t0 = time.time()
for _ in range(reps):
    ev.rank_all(test, n)
torch.cuda.synchronize(); dt = (time.time() - t0) / reps
print(f"eval {name} n={n}: {dt*1e3:.3f} ms -> {n/dt:.0f} test triples/s")
