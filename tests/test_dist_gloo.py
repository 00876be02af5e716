"""N>1 path on CPU: world_size-2 gloo run of the data-parallel Trainer (batch sharded across ranks; per step the flat
gradient is reduce-scattered, each rank runs the dense optimiser on ITS 1/N shard of the tables with 1/N of the
optimiser state, and the updated parameters are all-gathered) must reproduce the single-process full-batch result.  The compute backend is the test-only oracle backend (tests/oracle_backend.py);
what is under test is the sharding / collective / replica-consistency logic of pykg2vec_amd.trainer+generator."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, model_name, opt, out_dir, sparse=None, tag="", allreduce=None):
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_backend
    import hip_util
    from golden_util import Case
    from pykg2vec_amd.trainer import Trainer
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    c = Case(model_name)
    cfg = hip_util.make_config(c.E, c.R, c.hp, c.train, c.valid, c.test, optimizer=opt, lr=0.05, batch_size=64,
                               device="cpu")
    cfg.debug = False
    cfg.tot_train_triples = 64 * 3  # three steps per epoch
    m = hip_util.model_from_case(c, device="cpu")
    tr = Trainer(m, cfg, backend=oracle_backend)
    if sparse is not None:
        tr.switches["dp_sparse"] = sparse
    if allreduce is not None:
        tr.switches["dp_allreduce"] = allreduce
    tr.build_model()
    tr.generator = tr._new_generator()
    losses = [tr.train_model_epoch(e) for e in range(2)]
    ranks = tr.evaluator.rank_all(c.test, 8).numpy()
    assert tr._sparse_dp == bool(sparse and world > 1)
    if world > 1 and not tr._sparse_dp:
        assert tr._dp_allreduce == (allreduce is not False)     # (default at these table sizes: one all-reduce)
    if tr._sparse_dp or tr._dp_allreduce:   # the optimiser (and its state) is replicated, nothing is sharded
        assert tr.flat.param_shard.numel() == tr.flat.numel and (tr.flat.state1 is None or tr.flat.state1.numel() == tr.flat.numel)
    else:
        assert tr.flat.numel % (4 * world) == 0 and tr.flat.param_shard.numel() * world == tr.flat.numel
        if tr.flat.state1 is not None:  # optimiser state exists only for this rank's shard
            assert tr.flat.state1.numel() * world == tr.flat.numel
    np.savez(os.path.join(out_dir, "r%d_w%d%s.npz" % (rank, world, tag)), losses=np.asarray(losses), ranks=ranks,
             **{n: p.detach().numpy() for n, p in m.named_parameters()})
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("allreduce", [False, None], ids=["sharded", "allreduce"])
@pytest.mark.parametrize("model_name,opt", [("transe_l1", "adam"), ("distmult", "adagrad"), ("rotate", "sgd"),
                                            ("complex", "adagrad"), ("rescal", "adam"), ("transh_l2", "rms")])
def test_two_rank_data_parallel_equals_single_process(tmp_path, model_name, opt, allreduce):
    out = str(tmp_path)
    _run(0, 1, 0, model_name, opt, out)
    port = _free_port()
    # sharded = reduce-scatter -> optimiser on the rank's shard -> all-gather (tables beyond 32 MB); the default at these sizes is
    # one all-reduce with the optimiser replicated (Trainer._dp_allreduce_wanted)
    mp.spawn(_run, args=(2, port, model_name, opt, out, None, "", allreduce), nprocs=2, join=True)
    one = np.load(os.path.join(out, "r0_w1.npz"))
    a = np.load(os.path.join(out, "r0_w2.npz"))
    b = np.load(os.path.join(out, "r1_w2.npz"))
    for k in one.files:
        if k in ("losses", "ranks"):
            continue
        assert np.array_equal(a[k], b[k]), "replicas diverged on %s" % k            # bit-identical replicas
        assert np.allclose(a[k], one[k], atol=2e-5, rtol=1e-4), (k, np.abs(a[k] - one[k]).max())
    assert np.allclose(a["losses"], one["losses"], rtol=1e-4)
    assert np.array_equal(a["ranks"], b["ranks"])
    assert np.abs(a["ranks"] - one["ranks"]).max() <= 1


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("model_name,opt,allreduce", [("transe_l1", "adam", False), ("transe_l1", "adam", None), ("rotate", "sgd", None),
                                                      ("rescal", "adam", False), ("complex", "adagrad", None)])
def test_four_and_eight_rank_data_parallel_equals_single_process(tmp_path, world, model_name, opt, allreduce):
    """The node the north star names has 8 GPUs: the same sharding / exchange / replica-consistency logic at world sizes 4 and 8
    (a batch of 64 positives leaves 16 / 8 per rank; the flat buffers are padded to a multiple of 4 * world floats)."""
    out = str(tmp_path)
    _run(0, 1, 0, model_name, opt, out)
    mp.spawn(_run, args=(world, _free_port(), model_name, opt, out, None, "", allreduce), nprocs=world, join=True)
    one = np.load(os.path.join(out, "r0_w1.npz"))
    reps = [np.load(os.path.join(out, "r%d_w%d.npz" % (r, world))) for r in range(world)]
    for k in one.files:
        for b in reps[1:]:
            assert np.array_equal(reps[0][k], b[k]), "replicas diverged on %s" % k
        if k not in ("losses", "ranks"):
            assert np.allclose(reps[0][k], one[k], atol=2e-5, rtol=1e-4), (k, np.abs(reps[0][k] - one[k]).max())
    assert np.allclose(reps[0]["losses"], one["losses"], rtol=1e-4)
    assert np.abs(reps[0]["ranks"] - one["ranks"]).max() <= 1


@pytest.mark.parametrize("model_name,opt", [("rescal", "adam"), ("rescal", "sgd"), ("rescal", "rms")])
def test_sparse_row_exchange_equals_the_dense_exchange_bit_for_bit(tmp_path, model_name, opt):
    """Two ranks, gradient rows of the touched entities exchanged as lists (Trainer._sparse_exchange) instead of the dense
    reduce-scatter / all-gather of the flat buffers: identical tables on both ranks, and identical -- np.array_equal -- to the dense
    exchange's (at world size 2 a row's sum is one addition either way)."""
    out = str(tmp_path)
    port = _free_port()
    mp.spawn(_run, args=(2, port, model_name, opt, out, False, "_dense"), nprocs=2, join=True)
    port = _free_port()
    mp.spawn(_run, args=(2, port, model_name, opt, out, True, "_sparse"), nprocs=2, join=True)
    d0 = np.load(os.path.join(out, "r0_w2_dense.npz"))
    s0 = np.load(os.path.join(out, "r0_w2_sparse.npz"))
    s1 = np.load(os.path.join(out, "r1_w2_sparse.npz"))
    for k in d0.files:
        assert np.array_equal(s0[k], s1[k]), "replicas diverged on %s" % k
        assert np.array_equal(s0[k], d0[k]), (k, np.abs(s0[k] - d0[k]).max())


def _run_square(rank, world, port, model_name, out_dir, sparse, tag):
    """tot_relation == tot_entity: the relation tables have as many rows as the entity table (the sparse-row exchange
    used to pick its tables by row count and would have reduced the relation tables only at the rows named by ENTITY ids)."""
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_backend
    import hip_util
    import kge_oracle as ko
    from pykg2vec_amd.trainer import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = 40                                       # entities == relations
    rng = np.random.default_rng(17)
    hp = dict(hidden_size=6, margin=1.0) if model_name == "rescal" else dict(ent_hidden_size=6, rel_hidden_size=5, margin=1.0, l1_flag=True)
    trip = np.stack([rng.integers(N, size=160), rng.integers(N, size=160), rng.integers(N, size=160)], 1).astype(np.int64)
    P = ko.init_params(model_name, rng, tot_entity=N, tot_relation=N, **{k: v for k, v in hp.items() if k.endswith("hidden_size")})
    cfg = hip_util.make_config(N, N, hp, trip[:128], trip[128:144], trip[144:], optimizer="adam", lr=0.05, batch_size=8, device="cpu")
    cfg.tot_train_triples = 8 * 4
    m = hip_util.model_from_params(model_name, P, hp, N, N, device="cpu")
    tr = Trainer(m, cfg, backend=oracle_backend)
    tr.switches["dp_sparse"] = sparse
    tr.build_model()
    tr.generator = tr._new_generator()
    for e in range(2):
        tr.train_model_epoch(e)
    assert tr._sparse_dp == sparse
    np.savez(os.path.join(out_dir, "sq_r%d%s.npz" % (rank, tag)), **{n: p.detach().numpy() for n, p in m.named_parameters()})
    dist.destroy_process_group()


@pytest.mark.parametrize("model_name", ["rescal", "transr"])
def test_sparse_row_exchange_with_as_many_relations_as_entities(tmp_path, model_name):
    out = str(tmp_path)
    mp.spawn(_run_square, args=(2, _free_port(), model_name, out, False, "_dense"), nprocs=2, join=True)
    mp.spawn(_run_square, args=(2, _free_port(), model_name, out, True, "_sparse"), nprocs=2, join=True)
    d0, s0, s1 = (np.load(os.path.join(out, "sq_r%d_%s.npz" % (r, t))) for r, t in ((0, "dense"), (0, "sparse"), (1, "sparse")))
    for k in d0.files:
        assert np.array_equal(s0[k], s1[k]), "replicas diverged on %s" % k
        assert np.array_equal(s0[k], d0[k]), (k, np.abs(s0[k] - d0[k]).max())
