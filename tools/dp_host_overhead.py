"""Dev tool: is the data-parallel C1 step host-bound?  One rank, but with a ONE-rank RCCL process group handed to the Trainer, so the step
takes the N > 1 code path (owner-computes gradient -> all-reduce / reduce-scatter -> optimiser -> row norms, eager launches from
Python): wall time per step against the sum of the kernels' own durations says how much of an N > 1 step would be host time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import bench
from pykg2vec_amd.trainer import Trainer
import pykg2vec_amd.pairwise as pw

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
for allred in ("1", "0"):
    os.environ["KGE_DP_ALLREDUCE"] = allred
    os.environ["KGE_PULL"] = "1"
    train, valid, test = bench.synthetic_split(bench.E, bench.R, (bench.N_TRAIN, 1000, 1000))
    cfg = bench.make_config(bench.E, bench.R, bench.N_TRAIN, 32768, "cuda:0", hidden_size=100, l1_flag=True)
    cfg.knowledge_graph = bench._KG({"triplets_train": train, "triplets_valid": valid, "triplets_test": test, "hr_t": {}, "tr_h": {}})
    torch.manual_seed(0)
    m = pw.TransE(**cfg.__dict__)
    tr = Trainer(m, cfg, process_group=dist.group.WORLD, use_graph=False)
    tr.build_model()
    tr.generator = tr._new_generator()
    assert tr._pull_dp_ok(), "not on the data-parallel owner-computes path"
    spe = bench.N_TRAIN // 32768
    def run(n):
        while n > 0:
            if tr.generator._pending <= 0:
                tr.generator.start_one_epoch(spe)
            k = min(n, tr.generator._pending)
            tr.step_next_batches(k)
            n -= k
    run(2 * spe); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200); t_host = time.perf_counter() - t0   # host time to ENQUEUE 200 steps
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print("KGE_DP_ALLREDUCE=%s: %.1f us per step wall (200 steps), host enqueue %.1f us per step" % (allred, t_all / 200 * 1e6, t_host / 200 * 1e6), flush=True)
    del tr, m
dist.destroy_process_group()
