"""N>1 on real kernels: two ranks sharing ONE GPU over gloo run the data-parallel TransE / TransM trainer with the
owner-computes gradient step (kge_pull_step in KGE_OPT_GRADIENT mode on the rank's share of each batch, then reduce-scatter /
sharded optimiser / all-gather / row norms) and must (a) end with bit-identical replicas and (b) reproduce the single-process
run on the same global batches: the ranks' shares of the Philox stream -- including the sampler that rides ahead in the
previous step's launch -- have to add up to exactly the single-process negatives."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, model, opt, l1, out_dir):
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["KGE_PULL"] = "1"
    import hip_util
    import kge_oracle as ko
    from pykg2vec_amd.trainer import Trainer
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    E, R, D, B = 700, 23, 32, 256
    n_train = 3 * B + 40
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    hp = dict(hidden_size=D, l1_flag=l1, margin=1.0)
    P = ko.init_params("transe", rng, tot_entity=E, tot_relation=R, hidden_size=D)
    cfg = hip_util.make_config(E, R, hp, train, train[:4], train[:16], optimizer=opt, lr=0.01, batch_size=B)
    m = hip_util.model_from_params(model, P, hp, E, R, train=train)
    tr = Trainer(m, cfg, use_graph=False)
    tr.build_model()
    tr.generator = tr._new_generator()
    assert (tr._pull_dp_ok() if world > 1 else tr._pull_ok())
    losses = [tr.train_model_epoch(e) for e in range(3)]       # 3 epochs x 3 full batches: the ride-along sampler crosses epochs
    np.savez(os.path.join(out_dir, "r%d_w%d.npz" % (rank, world)), losses=np.asarray(losses),
             **{n: p.detach().cpu().numpy() for n, p in m.named_parameters()})
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,opt,l1", [("transe", "adam", True), ("transe", "sgd", False), ("transm", "adagrad", True)])
def test_two_ranks_owner_computes_gradient_equals_single_process(tmp_path, model, opt, l1):
    out = str(tmp_path)
    _run(0, 1, 0, model, opt, l1, out)
    mp.spawn(_run, args=(2, _free_port(), model, opt, l1, out), nprocs=2, join=True)
    one = np.load(os.path.join(out, "r0_w1.npz"))
    a = np.load(os.path.join(out, "r0_w2.npz"))
    b = np.load(os.path.join(out, "r1_w2.npz"))
    for k in one.files:
        if k == "losses":
            continue
        assert np.array_equal(a[k], b[k]), "replicas diverged on %s" % k
        bad = ~np.isclose(a[k], one[k], atol=2e-5, rtol=1e-4)
        assert bad.mean() <= (0.0 if opt == "sgd" else 2e-3), (k, bad.mean(), np.abs(a[k] - one[k]).max())
    assert np.allclose(a["losses"], one["losses"], rtol=1e-4), (a["losses"], one["losses"])


# ---------------------------------------------------------------------------------------------------------------------
# The RCCL branch on the one GPU a test box has: a ONE-rank "nccl" process group handed to the Trainer explicitly makes the
# step run reduce_scatter_tensor (SUM for the hinge, AVG for the mean-type losses) -> sharded optimiser ->
# all_gather_into_tensor exactly as at N > 1 (degenerate collectives, same calls), eagerly and captured in a hipGraph
# (KGE_GRAPH_MULTI=1), and with the owner-computes gradient step in front (KGE_PULL=1).  Each must reproduce the plain
# single-process run.
def _run_rccl_one_rank(rank, port, case, out_dir):
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    model, opt, mode = case
    for k in ("KGE_PULL", "KGE_GRAPH_MULTI", "KGE_STAGED", "KGE_DP_SPARSE"):
        os.environ.pop(k, None)
    if mode == "graph":
        os.environ["KGE_GRAPH_MULTI"] = "1"
    if mode == "sparse":       # gradient rows of the batch's entities exchanged as lists, replicated optimiser (Trainer._sparse_exchange)
        os.environ["KGE_DP_SPARSE"] = "1"
    # tables of this size take ONE all-reduce by default (Trainer._dp_allreduce_wanted); the other cases force the sharded
    # reduce-scatter / all-gather step that larger tables use
    allreduce = mode.endswith("allreduce")
    os.environ["KGE_DP_ALLREDUCE"] = "1" if allreduce else "0"
    mode = mode.replace("+allreduce", "").replace("allreduce", "eager")
    os.environ["KGE_PULL"] = "1" if mode == "pull" else "0"
    import hip_util
    import kge_oracle as ko
    from pykg2vec_amd.trainer import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    rng = np.random.default_rng(11)
    E, R, D, B = 600, 19, 32, 256
    n_train = 4 * B
    train = np.stack([rng.integers(E, size=n_train), rng.integers(R, size=n_train), rng.integers(E, size=n_train)], 1)
    hp = (dict(hidden_size=D, l1_flag=True, margin=1.0) if model == "transe" else
          dict(hidden_size=D, margin=1.0) if model == "rescal" else dict(hidden_size=D, lmbda=1e-3))
    P = ko.init_params(model, rng, tot_entity=E, tot_relation=R, hidden_size=D)
    out = {}
    for label, pg in (("plain", None), ("rccl", dist.group.WORLD)):
        cfg = hip_util.make_config(E, R, hp, train, train[:4], train[:16], optimizer=opt, lr=0.01, batch_size=B)
        m = hip_util.model_from_params(model, P, hp, E, R, train=train)
        if pg is None:     # the plain run must not see the default group
            tr = Trainer(m, cfg, use_graph=False)
            tr.distributed, tr.world_size, tr.rank = False, 1, 0
        else:
            tr = Trainer(m, cfg, process_group=pg)
        tr.build_model()
        tr.generator = tr._new_generator()
        if pg is not None:
            assert tr.distributed and tr.world_size == 1 and tr._collectives() == (True, "nccl")
            assert tr._sparse_dp == (mode == "sparse") and tr._dp_allreduce == allreduce
            assert (tr.flat.grad_shard.data_ptr() != tr.flat.grad.data_ptr()) == (mode != "sparse" and not allreduce)
            assert tr._graph_wanted(4) == (mode == "graph")
            assert tr._pull_dp_ok() == (mode == "pull")
        losses = [tr.train_model_epoch(e) for e in range(2)]
        if pg is not None and mode == "graph":
            assert tr._graph is not None
        torch.cuda.synchronize()
        out[label] = (losses, {n: p.detach().cpu().numpy() for n, p in m.named_parameters()})
        if pg is not None and mode != "graph":   # the per-phase events bench.py reads at N > 1
            tr.phase_marks = []
            tr.train_model_epoch(2)
            torch.cuda.synchronize()
            names = [n for n, _ in tr.phase_marks]
            times = [a[1].elapsed_time(b[1]) for a, b in zip(tr.phase_marks, tr.phase_marks[1:])]
            assert min(times) >= 0.0
            tr.phase_marks = None
            out["marks"] = names
    np.savez(os.path.join(out_dir, "rccl1.npz"), marks=np.asarray(out.get("marks", []), dtype="U32"), plain_losses=np.asarray(out["plain"][0]), rccl_losses=np.asarray(out["rccl"][0]),
             **{"plain." + k: v for k, v in out["plain"][1].items()}, **{"rccl." + k: v for k, v in out["rccl"][1].items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [("transe", "adam", "eager"), ("transe", "adam", "graph"), ("transe", "sgd", "pull"),
                                  ("complex", "adagrad", "eager"), ("complex", "adagrad", "graph"),
                                  ("rescal", "adam", "eager"),    # separate sampler launch: it runs under the async all-gather
                                  ("rescal", "adam", "sparse"),
                                  ("transe", "adam", "allreduce"), ("transe", "sgd", "pull+allreduce"), ("complex", "adagrad", "allreduce")],
                         ids=lambda c: "-".join(c))
def test_one_rank_rccl_group_runs_the_collective_step(tmp_path, case):
    out = str(tmp_path)
    mp.spawn(_run_rccl_one_rank, args=(_free_port(), case, out), nprocs=1, join=True)
    z = np.load(os.path.join(out, "rccl1.npz"))
    assert np.allclose(z["plain_losses"], z["rccl_losses"], rtol=1e-4), (z["plain_losses"], z["rccl_losses"])
    marks = [str(x) for x in z["marks"]]
    if case[2].endswith("allreduce"):   # one collective per step, no parameter all-gather
        want = ["begin", "compute", "reduce_scatter", "optimiser"] + (["row_norms"] if "pull" in case[2] else [])
        assert marks == want * 4, marks
    elif case[2] == "pull":      # four full batches through the owner-computes gradient step
        assert marks == ["begin", "compute", "reduce_scatter", "optimiser", "all_gather", "row_norms"] * 4, marks
    elif case[2] == "eager":
        assert marks == ["begin", "compute", "reduce_scatter", "optimiser", "all_gather"] * 4, marks
    elif case[2] == "sparse":  # no parameter all-gather: every rank steps every row
        assert marks == ["begin", "compute", "reduce_scatter", "optimiser"] * 4, marks
    for k in z.files:
        if not k.startswith("plain."):
            continue
        a, b = z[k], z["rccl." + k[6:]]
        bad = ~np.isclose(a, b, atol=2e-5, rtol=1e-4)   # (Adam / Adagrad first steps on atomically summed gradients move an
        # entry by ~lr * sign(g): isolated sign flips of rounding residues are allowed, nothing else)
        assert bad.mean() <= (0.0 if case[1] == "sgd" else 2e-3), (k, bad.mean(), np.abs(a - b).max())
