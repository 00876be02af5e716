"""Dev tool: raw kernel timings of the TransE FB15k-shape train step and rank sweep (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import hip_util
from pykg2vec_amd import kernels as K
from pykg2vec_amd.trainer import Trainer
from pykg2vec_amd.evaluator import Evaluator

E, R, NTRAIN, D = 14951, 1345, 483142, int(os.environ.get("D", 100))
rng = np.random.default_rng(1234)
train = np.stack([rng.integers(E, size=NTRAIN), rng.integers(R, size=NTRAIN), rng.integers(E, size=NTRAIN)], 1)
test = np.stack([rng.integers(E, size=8192), rng.integers(R, size=8192), rng.integers(E, size=8192)], 1)
hp = dict(hidden_size=D, l1_flag=True, margin=1.0)

def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

for opt in ("adam", "sgd"):
    for B in (128, 1024, 4096, 32768, 131072):
        cfg = hip_util.make_config(E, R, hp, train, test[:10], test, optimizer=opt, lr=0.01, batch_size=B)
        torch.manual_seed(0)
        m = hip_util.model_from_params("transe", {}, hp, E, R)
        tr = Trainer(m, cfg); tr.build_model()
        tr.generator = tr._new_generator()
        tr.generator.start_one_epoch(10**9)
        batch = next(tr.generator)
        t_gen = timeit(lambda: next(tr.generator))
        t_fused = timeit(lambda: tr._accumulate_pairwise(*batch))
        t_opt = timeit(lambda: tr.flat.optimizer_step(0.01))
        def step():
            tr._accumulate_pairwise(*next(tr.generator)); tr._reduce_and_step()
        t_step = timeit(step)
        alg = 2 * B * 3628
        print(f"opt={opt} B={B}: gen {t_gen:.1f}us fused {t_fused:.1f}us ({alg/t_fused/1e6:.2f} TB/s alg) opt {t_opt:.1f}us step {t_step:.1f}us -> {2*B/t_step:.1f} M scored triples/s", flush=True)
        if opt == "adam" and B == 32768:
            with torch.no_grad():
                pos = m(batch[0], batch[1], batch[2])
            t_fwd = timeit(lambda: K.score_forward(tr._desc, batch[0], batch[1], batch[2]))
            print(f"   fwd-only {t_fwd:.1f}us ({B*1228/t_fwd/1e6:.2f} TB/s alg)")
            ds = torch.ones(B, device="cuda")
            t_bwd = timeit(lambda: K.score_backward(tr._desc, batch[0], batch[1], batch[2], ds))
            print(f"   bwd-only {t_bwd:.1f}us")
# eval
cfg = hip_util.make_config(E, R, hp, train, test[:10], test)
m = hip_util.model_from_params("transe", {}, hp, E, R)
ev = Evaluator(m, cfg)
for n in (64, 1024, 8192):
    ev.rank_all(test, n); torch.cuda.synchronize()
    t0 = time.time(); ev.rank_all(test, n); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"eval n={n}: {dt*1e3:.2f} ms -> {n/dt:.0f} test triples/s ; {2*n*E*D*4/dt/1e12:.2f} TB/s alg ; {2*n*E/dt/1e9:.1f} G cand/s")
