"""Dev tool: run only the FB15k-shape TransE rank sweep (for rocprofv3 counter passes)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hip_util
from pykg2vec_amd.evaluator import Evaluator
E, R, D = 14951, 1345, 100
n = int(os.environ.get("N", 8192))
rng = np.random.default_rng(1234)
test = np.stack([rng.integers(E, size=n), rng.integers(R, size=n), rng.integers(E, size=n)], 1)
hp = dict(hidden_size=D, l1_flag=bool(int(os.environ.get("L1", 1))), margin=1.0)
cfg = hip_util.make_config(E, R, hp, test[:10], test[:10], test)
m = hip_util.model_from_params(os.environ.get("MODEL", "transe"), {}, hp, E, R)
ev = Evaluator(m, cfg)
for _ in range(3):
    ev.rank_all(test, n)
torch.cuda.synchronize()
t0 = time.time(); ev.rank_all(test, n); torch.cuda.synchronize(); dt = time.time() - t0
print(f"eval n={n}: {dt*1e3:.2f} ms -> {n/dt:.0f} test triples/s")
